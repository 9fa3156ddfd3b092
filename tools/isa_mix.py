import re, sys
from collections import Counter
path, pat = sys.argv[1], sys.argv[2]
s=open(path).read()
names=re.findall(r"^(_Z[A-Za-z0-9_]*):", s, re.M)
name=[n for n in names if pat in n][0]
i0=s.index("\n"+name+":"); i1=s.index("s_endpgm", i0)
body=s[i0:i1].split('\n')
print(name, len(body),"lines")
labels={}
for i,l in enumerate(body):
    mm=re.match(r"^(\.LBB\d+_\d+):",l)
    if mm: labels[mm.group(1)]=i
loops=[]
for i,l in enumerate(body):
    mm=re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)",l)
    if mm and mm.group(1) in labels and labels[mm.group(1)]<i:
        loops.append((labels[mm.group(1)],i))
a,b=max(loops,key=lambda x:x[1]-x[0])
print("loop lines",a,b)
loop=body[a:b+1]
c=Counter()
for l in loop:
    l=l.strip()
    if not l or l.startswith(('.',';','//')) : continue
    op=l.split()[0]
    if op.startswith('v_mfma'): c['mfma']+=1
    elif op.startswith('v_'): c['valu']+=1; c['v:'+op]+=1
    elif op.startswith('ds_'): c['lds']+=1; c['d:'+op]+=1
    elif op.startswith('s_'): c['salu']+=1; c['s:'+op]+=1
    elif op.startswith(('global_','buffer_','flat_','scratch_')): c['vmem']+=1
print({k:v for k,v in c.items() if ':' not in k})
print(sorted([(v,k) for k,v in c.items() if k.startswith('v:')],reverse=True)[:40])
print(sorted([(v,k) for k,v in c.items() if k.startswith('d:')],reverse=True))
print(sorted([(v,k) for k,v in c.items() if k.startswith('s:')],reverse=True)[:8])
