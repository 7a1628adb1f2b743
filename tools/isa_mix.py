"""Instruction mix of the kernels in a gfx950 .s file (hipcc -save-temps): whole kernel, and the blocks of its loops.
    python tools/isa_mix.py file.s [name-filter] [-v]"""
import sys, re, collections
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('-') else ""
verbose = '-v' in sys.argv
ks = re.split(r'\n(_Z[^\n:]*):[^\n]*\n', s)
def summ(tag, b):
    c = collections.Counter(x.split()[0] for x in b)
    g = lambda f: sum(v for k, v in c.items() if f(k))
    print("  %-26s n=%5d mfma=%4d valu=%4d ds_read=%3d ds_write=%3d vmem=%3d scratch=%3d lane=%3d accvgpr=%3d salu=%4d waitcnt=%3d nop=%3d branch=%3d" % (
        tag, len(b), g(lambda k: k.startswith('v_mfma')), g(lambda k: k.startswith('v_') and not k.startswith('v_mfma') and 'lane' not in k and not k.startswith('v_accvgpr')),
        g(lambda k: k.startswith('ds_read')), g(lambda k: k.startswith('ds_write')), g(lambda k: k.startswith('buffer_') or k.startswith('global_')),
        g(lambda k: k.startswith('scratch')), g(lambda k: 'readlane' in k or 'writelane' in k), g(lambda k: k.startswith('v_accvgpr')),
        g(lambda k: k.startswith('s_') and not k.startswith('s_waitcnt') and not k.startswith('s_nop') and not k.startswith('s_cbranch') and not k.startswith('s_branch')),
        c['s_waitcnt'], c['s_nop'], g(lambda k: k.startswith('s_cbranch') or k.startswith('s_branch'))))
for i in range(1, len(ks), 2):
    name = ks[i]
    if flt not in name: continue
    body = ks[i + 1].split('.Lfunc_end')[0]
    blocks = []; cur = []; label = 'entry'; inloop = False
    for l in body.split('\n'):
        ls = l.strip()
        if re.match(r'^\.LBB\d+_\d+:', ls):
            blocks.append((label, inloop, cur)); cur = []; label = ls.split(':')[0]; inloop = 'Loop' in ls
        elif ls.startswith('; %bb.') :
            if 'Loop' in ls: inloop = True
        elif ls and not ls.startswith(';') and not ls.startswith('.'):
            cur.append(ls)
    blocks.append((label, inloop, cur))
    print(name)
    summ('whole kernel', [x for _, _, b in blocks for x in b])
    summ('blocks inside loops', [x for _, il, b in blocks if il for x in b])
    if verbose:
        for lab, il, b in blocks:
            if len(b) > 20: summ(lab + (' (loop)' if il else ''), b)
