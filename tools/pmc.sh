# usage: bash tools/pmc.sh <tag> "<COUNTER list, one rocprofv3 pass each>"
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
for c in "$@"; do
  rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o $(echo $c | tr ' ' '_') -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
done
python3 - <<'PY'
import csv,glob,os,collections
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_'+os.environ.get('TAG','')
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_*/*counter_collection.csv')):
    acc=collections.defaultdict(lambda:[0,0])
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'sketch_filter' in k or 'verify_count' in k:
            key=(k.split('(')[0][-40:], r['Counter_Name'])
            acc[key][0]+=float(r['Counter_Value']); acc[key][1]+=1
    for (k,c),(v,n) in sorted(acc.items()):
        print(f"{k:42s} {c:28s} per-launch {v/n:16.0f}  (n={n})")
PY
