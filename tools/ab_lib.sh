for i in 1 2; do
UD_LIB_PATH=$PWD/unidefense_amd/libud_base.so timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('base', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('new ', d['value'], d['ms_per_step'])"
done
