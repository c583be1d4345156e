#!/usr/bin/env python3
"""Kernel resource table + instruction histogram from a hipcc -save-temps gfx950 .s file."""
import re
import sys
from collections import Counter

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else None
# metadata
for m in re.finditer(r'- \.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)', txt, re.S):
    name = re.sub(r'_ZN4plfx\d*', '', m.group(2))[:40]
    print('%-42s vgpr %3s agpr %3s sgpr %3s scratch %4s' % (name, m.group(5), m.group(1), m.group(4), m.group(3)))
if pat:
    m = re.search(r'^(_ZN4plfx\S*%s\S*):\n(.*?)s_endpgm' % pat, txt, re.S | re.M)
    if m:
        body = m.group(2)
        ops = Counter(re.findall(r'^\s+([a-z_0-9]+)', body, re.M))
        print('\n%s: %d instructions' % (m.group(1)[:60], sum(ops.values())))
        for k, v in ops.most_common(40):
            print('  %-28s %6d' % (k, v))
