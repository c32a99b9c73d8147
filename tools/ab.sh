for i in 1 2 3; do
for V in "" B C; do
L=$PWD/taxator-tk_amd/librpa_hip${V:+_$V}.so
RPA_LIB_PATH=$L python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${V:-A}', round(d['ms_per_step'],2), round(d['align_ms_per_step'],2))"
done; done
