set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --no-cpu-baseline --no-roofline --no-fp32-leg --no-call-leg --precision fp16"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_grp -- python3 $B --steps 1 --warmup 0 --overlap 0 > $OUT/pmc_grp.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/pmc_grp/**/*counter_collection.csv',recursive=True)[0]
agg=collections.Counter(); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"][:48]; agg[k]+=float(r["Counter_Value"])*1024*2; n[k]+=1
for k,v in agg.most_common(8): print("%-50s %4d launches fetch %.1f GB"%(k,n[k],v/1e9))
PY
rm -rf $OUT/pmc_grp
