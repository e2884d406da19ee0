for rep in 1 2 3; do for v in a b; do python bench.py --no-cpu-baseline --steps 100 --pipeline-depth 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('spin_us=$v', round(d['ms_per_step'],4), d['path']['host_phase_marks_ms_last_step'])"; done; done
