# A/B of list-decoder builds plus their HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes): bash tools/run_gf.sh NAME [NAME ...]
mkdir -p gpurun_out/r3
python tools/wide_ab.py "" "$@" "" "$@" > gpurun_out/r3/ab_$1.txt 2>&1; grep "B= 65536\|B= 24576\|L= 32\|L=256" gpurun_out/r3/ab_$1.txt
cd /tmp && export TMPDIR=/tmp
for v in "" "$@"; do for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3/pmc_v_$v/$c -- python3 $GRAFT_REPO_ROOT/tools/scl_pmc3.py "$v" 65536 8 > /dev/null 2>&1; done; done
cd $GRAFT_REPO_ROOT
for v in "" "$@"; do echo "== variant '$v'"; for c in FETCH_SIZE WRITE_SIZE; do python tools/pmc_by_grid.py gpurun_out/r3/pmc_v_$v/$c es_scl_wide; done; done
