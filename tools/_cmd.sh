for v in "" _hi8 "" _hi8; do
RADMMM_LIB_PATH="$PWD/rad_mmm_amd/libradmmm_hip$v.so" python bench.py --dominant-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib$v', round(d['avg_launch_ms']*1e3,1), 'us')"
done
