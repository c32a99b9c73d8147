for i in 1 2; do
for FB in auto 1 2; do
if [ $FB = auto ]; then unset RPA_FORCE_BIN; else export RPA_FORCE_BIN=$FB; fi
python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('force_bin=$FB', round(d['ms_per_step'],2), round(d['align_ms_per_step'],2), '%.3g'%d['roofline']['cell_updates_per_s_in_kernel'])"
done; done
