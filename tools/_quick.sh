timeout 800 python -m pytest tests -q -m gpu -x -k "golden or work_is_cut or task_queue or small_arena or zero_type or fresh or engine_settings or conformer_counts or sixteen_pockets_one or reference" 2>&1 | tail -2
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-serial-leg 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], list(d['roofline']['kernel_ms_per_launch'].values()), d['work']['wave_time_share'], d['work']['table_items_per_ligand_conformer'])"
python bench.py --pockets 16 --ligands 200000 --steps 1 --warmup 1 --no-cpu-baseline --no-serial-leg 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('p16', d['value'])"
timeout 600 python tools/stress_shape.py 196 2>&1 | tail -1 | cut -c1-60
