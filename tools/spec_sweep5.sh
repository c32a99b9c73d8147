# speculation chunks re-swept after the band kernels made an alignment ~1.4x (config 2) / ~3x (config 5) cheaper
for C in c2 c5; do
for SP in "0 10" "0 6" "0 20" "10 10" "14 10" "16 6"; do
set -- $SP
RPA_SPEC_ADAPTIVE=1 RPA_SPEC1=$1 RPA_SPEC2=$2 python bench.py --config $C --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$C spec=$SP', 'ms/step', round(d['ms_per_step'],2), 'align', round(d['align_ms_per_step'],2), 'comp/dem', round(d['computed_over_demanded'],3), 'rounds', d['rounds_per_step'], 'value %.3g'%d['value'])"
done; done
