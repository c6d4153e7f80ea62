R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do
rm -rf $R/gpurun_out/prof_write$i
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write$i -- python3 $R/bench.py --no-cpu-baseline --no-order10 --steps 6 --warmup 1 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/prof_write$i/*/*counter_collection.csv')[0]
v=[float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'pipeline_kernel' in r['Kernel_Name']]
print('WRITE_SIZE KiB per launch:', [round(x) for x in v])
PY
done
