import re,sys
def analyze(path, name='_Z27conv_winograd36s_f32_kernelILi16ELb0ELi0ELi4ELb0EEv9Wino4ArgsiiiiPjPf'):
    s=open(path).read()
    k=s[s.index(name+':'):]
    k=k[:k.index('.Lfunc_end')]
    lines=k.split('\n')
    # main loop header: the label line followed by "=>This Loop Header: Depth=1"
    hdr=[i for i,l in enumerate(lines) if 'This Loop Header: Depth=1' in l][0]
    # the phase barrier: last s_barrier within an ASMSTART block after hdr that is followed by cbranch to a label before hdr
    bar=[i for i,l in enumerate(lines) if l.strip()=='s_barrier' and i>hdr]
    # hot path ends at first barrier after the 144 mfmas
    cnt=0; end=None
    for i in range(hdr,len(lines)):
        if 'v_mfma' in lines[i]: cnt+=1
        if cnt>=143 and lines[i].strip()=='s_barrier': end=i; break
    body=[l.strip() for l in lines[hdr:end+6] if l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.')]
    c=lambda p: sum(1 for l in body if l.startswith(p))
    return dict(n=len(body), mfma=c('v_mfma'), readlane=c('v_readlane'), writelane=c('v_writelane'), scratch=c('scratch_'), valu=sum(1 for l in body if l.startswith('v_') and not l.startswith('v_mfma')), salu=sum(1 for l in body if l.startswith('s_') and not l.startswith('s_waitcnt') and not l.startswith('s_nop')), waitcnt=c('s_waitcnt'), nop=c('s_nop'), ds=c('ds_'), vmem=c('buffer_'))
for p in sys.argv[1:]:
    print(p, analyze(p))
