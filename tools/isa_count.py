"""Count instructions per basic block of a kernel in a hipcc -S listing: python tools/isa_count.py k.s <name-substring>"""
import sys, re, collections
s = open(sys.argv[1]).read()
for f in re.split(r'\n(?=_Z[^\n]*:[^\n]*@)', s):
    name = f.split(':', 1)[0]
    if sys.argv[2] not in name: continue
    cur = 'entry'; blocks = collections.OrderedDict({cur: []})
    for l in f.split('\n')[1:]:
        l = l.strip()
        if not l or l.startswith(';'): continue
        if l.startswith('.LBB') and ':' in l: cur = l.split(':')[0]; blocks[cur] = []; continue
        if l.startswith('.') or l.endswith(':'): continue
        blocks[cur].append(l.split()[0])
    print(name[:70])
    for b, ins in blocks.items():
        c = collections.Counter(i.split('_')[0] for i in ins)
        if len(ins) > 4: print('  ', b, len(ins), dict(c))
    big = max(blocks.items(), key=lambda kv: len(kv[1]))
    print('   biggest', big[0], collections.Counter(big[1]).most_common(14))
    m = re.search(r'\.vgpr_count:\s+(\d+)', f)
