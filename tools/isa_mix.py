"""Instruction mix per kernel from a hipcc -save-temps gfx950 .s file (diagnostic)."""
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_ZN2vt\w+):[^\n]*\n(.*?)s_endpgm', s, flags=re.S | re.M):
    name, body = m.group(1), m.group(2)
    ops = {}
    for line in body.split('\n'):
        mm = re.match(r'\s+([a-z_0-9]+)\s', line)
        if mm:
            ops[mm.group(1)] = ops.get(mm.group(1), 0) + 1
    top = sorted(ops.items(), key=lambda kv: -kv[1])[:14]
    print(name[6:34], sum(ops.values()), top)
