# parallel tree search of the guessing decoders at 2048 shots per launch: shots admitted at a time (percent of the grid) x side branches of a tree in flight
for pct in ${PCTS:-25 50 75 100 150}; do for fl in ${FLIGHT:-6}; do
echo -n "shots_pct=$pct tree_inflight=$fl: "; SWD_GDG_SHOTS_PCT=$pct SWD_GDG_INFLIGHT=$fl timeout 120 python scripts/bench_configs.py 3small 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_launch'],1), round(d['windows_per_s']))"
done; done
