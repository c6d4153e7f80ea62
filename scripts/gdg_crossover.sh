for sh in 1024 4096 8192 16384; do
for mode in par ser; do
if [ $mode = par ]; then export SWD_GDG_PAR_MAX_SHOTS=1000000; unset SWD_GDG_SERIAL; else export SWD_GDG_SERIAL=1; fi
echo -n "shots=$sh $mode: "; SWD_GDG_SHOTS=$sh timeout 200 python scripts/bench_configs.py 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_launch'],1), round(d['windows_per_s']))"
done; done
