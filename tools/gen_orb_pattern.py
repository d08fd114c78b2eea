#!/usr/bin/env python3
"""One-off: transcribe the 256x4 ORB test-pair constants (data, not code) from the reference's
constant table (vido_slam/src/ORBextractor.cc:140-398) into include/vido_orb_pattern.h.
Runs only in the build container (needs /root/reference); the generated header is committed."""
import re, sys
src = open('/root/reference/vido_slam/src/ORBextractor.cc').read()
i = src.index('bit_pattern_31_[256*4]'); j = src.index('};', i)
body = src[i:j]; body = body[body.index('{') + 1:]
body = re.sub(r'/\*.*?\*/', '', body, flags=re.S); body = re.sub(r'//.*', '', body)
nums = [int(x) for x in re.findall(r'-?\d+', body)]
assert len(nums) == 1024
rows = [' ' + ' '.join('{%d,%d,%d,%d},' % tuple(nums[4*t:4*t+4]) for t in range(k, k+4)) for k in range(0, 256, 4)]
sys.stdout.write('\n'.join(rows) + '\n')
