import sys, ctypes as C
sys.path.insert(0, ".")
import hikari_jl_amd as hk
from hikari_jl_amd import scenes
ctx = hk.Context.get(0)
L = hk._lib.lib()
for leaf in (None, "3", "6", "8"):
    ctx.set_option("HK_BVH_LEAF", leaf)
    for name, mk in (("two_spheres", lambda: scenes.cornell_box(64, 64, objects="two_spheres")), ("sphere_box", lambda: scenes.cornell_box(64, 64)), ("sky", lambda: scenes.sky_scene(64, 64, env_res=16))):
        s, f, c = mk()
        sh = hk.scene_handle(ctx, s)
        a, b, d = C.c_int32(), C.c_int32(), C.c_int32()
        L.hk_scene_bvh_info(sh, C.byref(a), C.byref(b), C.byref(d))
        print("leaf", leaf, name, "nodes", a.value, "leaf tris", b.value, "depth", d.value)
