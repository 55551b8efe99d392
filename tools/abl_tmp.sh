cd $GRAFT_REPO_ROOT/pyflwdir_amd/csrc && make clean >/dev/null 2>&1; make -j8 DEVTOOLS=1 2>&1 | grep -E "error" -A5 | head -20; cd $GRAFT_REPO_ROOT
PFD_TILE_ABLATE=16 python bench.py --size 30000 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | grep -E '^\[k_tile<|phases_ms' | sed 's/.*"phases_ms"/phases_ms/' | cut -c1-200 | tail -3
