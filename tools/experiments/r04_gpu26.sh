for rep in 1 2; do for split in 1 2 4; do
LWKZG_SPLIT=$split python bench.py --no-cpu-baseline --no-extra-legs --steps 20 2>/dev/null | python -c "
import json,sys; l=json.loads(sys.stdin.read()); print('split $split', l['value'], l['ms_per_step'], l['roofline']['avg_launch_ms'], l['kernels_avg_ms'])"
done; done
