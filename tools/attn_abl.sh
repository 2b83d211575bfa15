for a in 0 1 2 3; do echo "== abl $a"; SRH_ATTN_ABL=$a python bench.py --no-cpu-baseline --no-check --steps 10 2>&1 | tail -1 | cut -c1-2500; done
