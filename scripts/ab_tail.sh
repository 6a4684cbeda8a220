for t in 0 3 4 5; do
RAMSES_AMD_MG_TAIL=$t python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tail $t:', d['vcycle']['ms_per_vcycle'], d['vcycle']['final_error'])"
done
