# instruction-cache counters of a bench workload's kernel (are the kernel's inlined copies of the walk / the phases of different
# workgroups fetch-bound?):   scripts/icache_counters.sh [workload] [extra bench.py arguments]
WL=${1:-headline}; shift $(( $# < 1 ? $# : 1 ))
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_ic_$WL
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_ic_$WL -- python3 $R/bench.py --workload $WL --no-cpu-baseline --no-order0 --no-stream --steps 3 --warmup 1 $* > $R/gpurun_out/pmc_ic_$WL.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob('$R/gpurun_out/pmc_ic_$WL/*/*counter_collection.csv')
if not f: print(open('$R/gpurun_out/pmc_ic_$WL.log').read()[-1500:]); raise SystemExit
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if 'pipeline_kernel' in r['Kernel_Name'] or 'bp4_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
out={k: round(sum(v)/len(v)) for k,v in agg.items()}
if out.get('SQC_ICACHE_REQ'): out['miss_rate']=round(out.get('SQC_ICACHE_MISSES',0)/out['SQC_ICACHE_REQ'],5)
print('$WL', out)
PY
