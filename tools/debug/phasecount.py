import re,sys
lines=open(sys.argv[1]).read().split('\n')
pat=sys.argv[2] if len(sys.argv)>2 else 'substep_kernel'
start=[i for i,l in enumerate(lines) if re.match(r'^_Z\w*'+pat+r'\w*:',l)]
i0=start[0]
end=next(i for i in range(i0,len(lines)) if 's_endpgm' in lines[i])
cur='pre'; stats={}; order=[]
for l in lines[i0:end]:
    m=re.search(r'; MI_MARK (\d+)',l)
    if m: cur='after%s'%m.group(1); continue
    t=l.strip()
    if not t or t.startswith(';') or t.startswith('.') or t.split(';')[0].strip().endswith(':'): continue
    d=stats.setdefault(cur,{'n':0,'st':0,'ld':0,'valu':0,'ds':0,'acc':0,'glob':0,'salu':0,'wait':0})
    if cur not in order: order.append(cur)
    d['n']+=1
    if t.startswith('scratch_store'): d['st']+=1
    elif t.startswith('scratch_load'): d['ld']+=1
    elif t.startswith('v_accvgpr'): d['acc']+=1
    elif t.startswith('v_'): d['valu']+=1
    elif t.startswith('ds_'): d['ds']+=1
    elif t.startswith('global_') or t.startswith('buffer_'): d['glob']+=1
    elif t.startswith('s_waitcnt'): d['wait']+=1
    elif t.startswith('s_'): d['salu']+=1
for k in order: print(k, stats[k])
