#!/bin/bash
# the default (three-stream) progressive encode: timeline of the last image -- when is the device idle?
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_prog3
rm -rf $out; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out -o prog --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --progressive --steps 4 --warmup 2 --no-cpu-baseline --no-psnr > $out/log.txt 2>&1
echo "profile rc=$?"; tail -1 $out/log.txt | cut -c1-200
python3 - $out <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/prog_timeline.txt
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if 'k_transform' in r['Kernel_Name']]
i0,i1=idx[-2],idx[-1]
t0=int(rows[i0]['Start_Timestamp'])
print("one image = %.1f us (transform to transform)" % ((int(rows[i1]['Start_Timestamp'])-t0)/1e3))
# idle = time with no kernel running
ev=[]
for r in rows[i0:i1]:
    ev.append((int(r['Start_Timestamp']),1)); ev.append((int(r['End_Timestamp']),-1))
ev.sort()
run=0; idle=0; last=t0; idle_list=[]
for t,d in ev:
    if run==0 and t>last: idle+=t-last; idle_list.append(((last-t0)/1e3,(t-last)/1e3))
    run+=d; last=max(last,t) if run==0 else last
    if run==0: last=t
print("idle total %.1f us; idle intervals > 10 us:" % (idle/1e3), [(round(a),round(b)) for a,b in idle_list if b>10])
for r in rows[i0:i1]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    if (e-s)>30000 or 'compact' in r['Kernel_Name'] or 'copyBuffer' in r['Kernel_Name']:
        print("%9.1f %8.1f q%s  %s" % ((s-t0)/1e3,(e-s)/1e3, r['Queue_Id'], r['Kernel_Name'].split('(')[0][:60]))
PY
