# same-box A/B of the shipped library against a variant: tools/prio_ab.sh <variant> ; prints kernel ms / roofline fraction
V=$1
for lib in "" $V; do
  if [ -z "$lib" ]; then unset FSMC_HIP_LIB; else export FSMC_HIP_LIB=$PWD/fastsmc_amd/variants/lib$lib.so; fi
  for args in "--workload c3 --pairs 131072" "--workload c3 --pairs 262144" "--workload c3 --pairs 196608" ""; do
    python bench.py $args --steps 2 --warmup 1 --cpu-pairs 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib:-base}', '$args', round(d['roofline']['kernel_ms'],1), round(d['roofline']['frac'],4), d['config']['ibd_records_per_step'])"
  done
done
