#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
HK_LIB_PATH=scratch/lib_dbg.so python - <<'PY' 2>&1 | tail -30
import sys
sys.path.insert(0, ".")
import hikari_jl_amd as hk
from hikari_jl_amd import scenes
s, film, cam = scenes.bomex_scene(1024, 1024)
vp = hk.VolPath(max_depth=32, samples=32)
vp(s, film, cam)
st = vp.stats()
print("collisions", st.shadow_collisions, st.track_collisions, "dda", st.shadow_dda_steps, st.track_dda_steps, "casts", st.rays_shadow)
PY
