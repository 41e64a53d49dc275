import sys
src=open(sys.argv[1]).read().split('\n')
name=sys.argv[2]
start=[i for i,l in enumerate(src) if l.startswith(name+':')][0]
end=[i for i,l in enumerate(src[start:],start) if l.startswith('.Lfunc_end')][0]
body=src[start:end]
print("lines",len(body))
ev=[]
for i,l in enumerate(body):
    t=l.strip()
    if t.startswith('scratch_store'): ev.append((i,'SS'))
    elif t.startswith('scratch_load'): ev.append((i,'SL'))
    elif t.startswith('s_barrier'): ev.append((i,'BAR'))
    elif t.startswith('v_mfma'): ev.append((i,'M'))
    elif 'global_load_lds' in t: ev.append((i,'DMA'))
    elif t.startswith('buffer_store'): ev.append((i,'ST'))
out=[];last=None;cnt=0;first=0
for i,k in ev:
    if k==last: cnt+=1
    else:
        if last: out.append(f"{last}x{cnt}@{first}")
        last=k;cnt=1;first=i
out.append(f"{last}x{cnt}@{first}")
print(' '.join(out))
