timeout 800 python -m pytest tests -q -m gpu -x -k "golden or work_is_cut or task_queue or small_arena or zero_type or fresh" 2>&1 | tail -2
for fl in 0 4096; do PMX_TREE_FLAGS=$fl python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-serial-leg 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], list(d['roofline']['kernel_ms_per_launch'].values()), d['work']['tree_frames_per_ligand'], d['work']['walker_passes_per_ligand'])"; done
