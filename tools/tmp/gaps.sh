cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/gaps; mkdir -p gpurun_out/gaps
KWS_BENCH_DTYPE=bf16 KWS_BENCH_BATCH=4096 timeout -k 10 200 rocprofv3 --kernel-trace -d gpurun_out/gaps --output-format csv -- python3 tools/bench_models.py resnet__res15 > gpurun_out/gaps.log 2>&1 || { tail -5 gpurun_out/gaps.log; exit 1; }
python3 - <<'PY'
import csv,glob,collections
rows=[]
for path in glob.glob('gpurun_out/gaps/**/*_kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0][-40:]))
rows.sort()
last=rows[-60:]
gaps=[(b[0]-a[1])/1e3 for a,b in zip(last,last[1:])]
print('span ms',(last[-1][1]-last[0][0])/1e6,'busy ms',sum(e-s for s,e,_ in last)/1e6, 'gaps us: mean %.1f max %.1f'%(sum(gaps)/len(gaps),max(gaps)))
d=collections.defaultdict(list)
for s,e,k in last: d[k].append((e-s)/1e6)
for k,v in d.items(): print(k, len(v), 'avg ms %.4f'%(sum(v)/len(v)))
PY
