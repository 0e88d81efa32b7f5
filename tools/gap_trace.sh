#!/bin/bash
# what runs between two steps: the untrimmed kernel trace of a short bench run, one steady-state step printed with every gap
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/gap_prof
rocprofv3 --kernel-trace --output-format csv -d /tmp/gap_prof -o g -- python3 $R/bench.py --steps 30 --warmup 5 --spinup-ms 100 --cpu-sample 0 --e2e 0 --no-checks "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/gap_prof/**/*kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
# the last 3 steps: find the last 4 sketch_filter launches
idx=[i for i,r in enumerate(rows) if 'sketch_filter' in r['Kernel_Name']]
for name,(a,b) in (("spin-up (no kernel timing)", (idx[40], idx[42])), ("timed region (kernel timing on)", (idx[-4], idx[-2]))):
    print("==", name)
    prev=None
    for r in rows[a-3:b+1]:
        s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
        gap=(s-prev)/1e3 if prev else 0
        print(f"gap {gap:7.2f} us | {(e-s)/1e3:8.2f} us  {r['Kernel_Name'][:90]}")
        prev=e
# step = filter start to next filter start
for name,sel in (("spin-up", idx[30:200]), ("timed", idx[-26:-1])):
    d=[(int(rows[j]['Start_Timestamp'])-int(rows[i]['Start_Timestamp']))/1e3 for i,j in zip(sel,sel[1:])]
    d.sort(); print(name, "filter start to filter start: median %.1f us over %d steps" % (d[len(d)//2], len(d)))
PY
