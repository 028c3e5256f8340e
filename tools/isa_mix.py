"""Static instruction mix of one kernel in a hipcc -S listing: python tools/isa_mix.py file.s substring"""
import collections, re, sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
for m in re.finditer(r'^(\S*' + re.escape(key) + r'\S*):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M):
    c = collections.Counter()
    for l in m.group(2).split('\n'):
        l = l.strip()
        if not l or l[0] in '.;' or l.endswith(':'):
            continue
        op = l.split()[0]
        kind = ('valu' if op.startswith('v_') else 'lds' if op.startswith('ds_') else
                'vmem' if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else
                'wait/barrier' if op.startswith(('s_waitcnt', 's_barrier')) else 'salu' if op.startswith('s_') else 'other')
        c[kind] += 1
    print(m.group(1)[:70], dict(c))
