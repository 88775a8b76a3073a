cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3q}; mkdir -p $O
DZN_LINKAGE_DEBUG=1 timeout 300 python scripts/bench_linkage.py 5000 20000 35790 > $O/linkage.txt 2>&1; cat $O/linkage.txt
cd /tmp && export TMPDIR=/tmp
cat > /tmp/one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from diarizen_amd import ops
from oracle.gen_golden import linkage_scale_case
e = linkage_scale_case(n=35790, dim=256, K=12, seed=5)
ops.linkage_centroid(e)
PY
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o lk -- python /tmp/one.py > $O/prof.log 2>&1
python - <<PY
import sqlite3, numpy as np, glob
f=glob.glob("$O/prof/*.db")[0]
c=sqlite3.connect(f)
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]; ks=[t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows=c.execute(f"select d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%step_kernel%' order by d.start").fetchall()
st=np.array(rows); d=(st[:,1]-st[:,0])/1e3
print("step launches", len(d), "mean us", d.mean(), "median", np.median(d), "hist", np.histogram(d, bins=[0,6,8,10,12,14,16,18,20,25,30,40,100])[0])
PY
rm -rf $O/prof
