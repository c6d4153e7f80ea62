# instruction-cache counters of the bench kernel (is the ~150 KB kernel, whose workgroups sit in different phases, fetch-bound?)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_ic
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_ic -- python3 $R/bench.py --no-cpu-baseline --no-order10 --steps 3 --warmup 1 > $R/gpurun_out/pmc_ic.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob('$R/gpurun_out/pmc_ic/*/*counter_collection.csv')
if not f: print(open('$R/gpurun_out/pmc_ic.log').read()[-1500:]); raise SystemExit
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if 'pipeline_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items(): print(k, round(sum(v)/len(v)))
PY
