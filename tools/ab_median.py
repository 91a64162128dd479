import re,statistics,sys
cur=None;d={}
for l in open(sys.argv[1]):
    l=l.strip()
    if l.startswith("== "): cur=l[3:]; d.setdefault(cur,[])
    m=re.search(r"launch \d+: ([0-9.]+) ms",l)
    if m and cur: d[cur].append(float(m.group(1)))
for k,v in d.items(): print(k,"median %.4f min %.4f mean %.4f n=%d"%(statistics.median(v),min(v),statistics.mean(v),len(v)))
