for C in c2 c4; do
for SP in "0 0" "0 4" "0 6" "0 10" "8 6" "6 4" "12 8"; do
set -- $SP
RPA_SPEC1=$1 RPA_SPEC2=$2 python bench.py --config $C --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$C spec=$SP', 'ms/step', round(d['ms_per_step'],2), 'align', round(d['align_ms_per_step'],2), 'comp/dem', round(d['computed_over_demanded'],3), 'rounds', d['rounds_per_step'], 'value %.3g'%d['value'])"
done; done
