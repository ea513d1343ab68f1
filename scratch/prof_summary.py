import sqlite3, glob, sys
db=sqlite3.connect(glob.glob(sys.argv[1]+'/*/*.db')[0]); cur=db.cursor()
rows=list(cur.execute("select name, start, end, grid_x from kernels order by start"))
# last call = after the last k_bbox_partial pair start
idx=[i for i,r in enumerate(rows) if 'k_bbox_partial' in r[0]]
start=idx[-2] if len(idx)>=2 else 0
t0=rows[start][1]; agg={}
for r in rows[start:]:
    nm=r[0].replace('void pcu::','').split('(')[0]
    agg.setdefault(nm,[0,0.0]); agg[nm][0]+=1; agg[nm][1]+=(r[2]-r[1])/1000
print("span us", (rows[-1][2]-t0)/1000)
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:25]: print(f"{k:50s} n={v[0]:4d} total={v[1]:9.1f} us")
