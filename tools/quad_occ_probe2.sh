# occupancy experiment of the quad step kernel: first LM steps of a full batch (tools/quad_probe.py) for builds with 1 / 2 waves per SIMD,
# with and without the tile aliased onto dead LDS (24 192 B: six waves per CU; WRONG results, timing only)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for cfg in "1 0" "2 0" "2 1"; do
  set -- $cfg
  export LIW_QUAD_OCC=$1
  if [ "$2" = "1" ]; then export LIW_QUAD_TILE_ALIAS=1; else unset LIW_QUAD_TILE_ALIAS; fi
  python -c "import __graft_entry__ as g; g.build()"
  echo "== OCC=$1 ALIAS=$2"; python tools/quad_probe.py 2>&1 | tail -1
done
