"""Instruction mix of the loops of a kernel in a hipcc -S listing: python loopstat.py file.s <mangled-name substring> ..."""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
for name in sys.argv[2:]:
    m0 = re.search(r'^(\S*' + re.escape(name) + r'\S*):', s, re.M)
    i = m0.start()
    j = s.index('.Lfunc_end', i)
    body = s[i:j].split('\n')
    labels = {}
    ins = []
    for l in body[1:]:
        t = l.strip()
        if not t or t.startswith(';'):
            continue
        if re.match(r'^\.LBB\d+_\d+:', t):
            labels[t.split(':')[0]] = len(ins)
            continue
        if t.startswith('.') or t.endswith(':'):
            continue
        ins.append(t)
    print(m0.group(1), 'instr', len(ins))
    for k, t in enumerate(ins):
        m = re.match(r's_cbranch\S*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', t)
        if m:
            lab = m.group(1) or m.group(2)
            if lab in labels and labels[lab] < k:
                a = labels[lab]
                c = Counter(x.split()[0] for x in ins[a:k + 1])
                print('  loop', lab, a, k, 'len', k - a + 1)
                grp = Counter()
                for op, v in c.items():
                    if op.startswith('v_mfma'): grp['mfma'] += v
                    elif op.startswith('ds_'): grp[op] += v
                    elif op.startswith('v_exp'): grp['v_exp'] += v
                    elif op.startswith('v_'): grp['valu'] += v
                    elif op.startswith('s_waitcnt'): grp['waitcnt'] += v
                    elif op.startswith('s_'): grp['salu'] += v
                    else: grp[op] += v
                print('   ', dict(grp))
                print('   ', [(o, v) for o, v in c.most_common(60) if o.startswith('v_') and not o.startswith('v_mfma')])
