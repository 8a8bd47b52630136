"""julia/HikariMI355X.jl cannot run here (no Julia, no Raycore), so what CAN be checked is checked: the `struct Hk*` mirrors in the
Julia file have exactly the C layout of include/hikari_mi355x.h (field names, order, offsets, total size — through the ctypes
mirror, which test_abi_and_host.py pins to gcc's sizeof), every `ccall` names a symbol the header declares with the number of
arguments the C prototype has, every Hk* constructor call passes one argument per field, and the shim has a method for
everything INTEGRATION.md says it does (`flatten_scene` was an empty declaration)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "julia", "HikariMI355X.jl")).read()

NAMES = {"HkTexture": "hk_texture", "HkTexRgba": "hk_tex_rgba", "HkTexF32": "hk_tex_f32", "HkMaterial": "hk_material", "HkPlSpectrum": "hk_pl_spectrum",
         "HkMediumInterface": "hk_medium_interface", "HkTriMeta": "hk_tri_meta", "HkLight": "hk_light", "HkEnvmap": "hk_envmap", "HkMedium": "hk_medium",
         "HkSceneDesc": "hk_scene_desc", "HkTables": "hk_tables", "HkIntegratorParams": "hk_integrator_params", "HkCamera": "hk_camera",
         "HkPostprocessParams": "hk_postprocess_params", "HkDenoiseParams": "hk_denoise_params"}
PRIM = {"Int32": (4, 4), "UInt32": (4, 4), "Float32": (4, 4), "Int64": (8, 8), "UInt8": (1, 1)}


def _structs():
    out = {}
    for m in re.finditer(r"^struct (Hk\w+)\n(.*?)^end", SRC, re.S | re.M):
        fields = []
        for line in m.group(2).splitlines():
            line = line.split("#")[0]
            for decl in line.split(";"):
                decl = decl.strip()
                if decl:
                    name, typ = decl.split("::")
                    fields.append((name.strip(), typ.strip()))
        out[m.group(1)] = fields
    return out


def _layout(typ, structs):
    """-> (size, align) of a Julia isbits type under the C layout rules Julia uses for them"""
    if typ in PRIM:
        return PRIM[typ]
    if typ.startswith("Ptr{"):
        return 8, 8
    m = re.match(r"NTuple\{(\d+),\s*(\w+)\}", typ)
    if m:
        s, a = _layout(m.group(2), structs)
        return int(m.group(1)) * s, a
    off, align = 0, 1
    for _, t in structs[typ]:
        s, a = _layout(t, structs)
        off = (off + a - 1) // a * a + s
        align = max(align, a)
    return (off + align - 1) // align * align, align


def test_julia_structs_have_the_c_layout(hk):
    structs = _structs()
    assert set(structs) == set(NAMES), set(structs) ^ set(NAMES)
    for jl, cname in NAMES.items():
        ct = getattr(hk._abi, cname)
        cfields = [f[0] for f in ct._fields_]
        assert [n for n, _ in structs[jl]] == cfields, (jl, [n for n, _ in structs[jl]], cfields)
        off = 0
        for name, typ in structs[jl]:
            s, a = _layout(typ, structs)
            off = (off + a - 1) // a * a
            assert off == getattr(ct, name).offset, (jl, name, off, getattr(ct, name).offset)
            assert s == getattr(ct, name).size, (jl, name, s, getattr(ct, name).size)
            off += s
        assert _layout(jl, structs)[0] == C.sizeof(ct), jl


def _split_top(s):
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return parts


def _call_args(text, start):
    """text[start] is the '(' of a call: -> list of top-level arguments"""
    depth, i = 0, start
    while True:
        ch = text[i]
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
            if depth == 0:
                break
        i += 1
    return _split_top(text[start + 1:i])


def test_every_ccall_matches_the_header(hk):
    hdr = open(os.path.join(ROOT, "include", "hikari_mi355x.h")).read()
    protos = {}
    for m in re.finditer(r"\b(?:int32_t|const char\*|void\*)\s+(hk_\w+)\s*\(([^;]*?)\);", hdr, re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len(_split_top(args))
    seen = set()
    for m in re.finditer(r"ccall\(\(:(hk_\w+), LIB\)", SRC):
        name = m.group(1)
        assert name in protos, name
        args = _call_args(SRC, m.start() + len("ccall"))
        argtypes = args[2].strip()
        assert argtypes.startswith("(") and argtypes.endswith(")")
        n_types = len([a for a in _split_top(argtypes[1:-1]) if a.strip()])
        assert n_types == protos[name], (name, n_types, protos[name])
        assert len(args) == 3 + n_types, (name, len(args), n_types)          # one value per declared argument type
        seen.add(name)
    need = {"hk_ctx_create", "hk_ctx_set_tables", "hk_scene_create", "hk_scene_destroy", "hk_integrator_create", "hk_film_create", "hk_film_clear", "hk_render",
            "hk_film_read_rgb", "hk_film_reduce", "hk_comm_create", "hk_comm_destroy", "hk_postprocess", "hk_denoise", "hk_film_fill_aux", "hk_ctx_destroy",
            "hk_film_destroy", "hk_integrator_destroy", "hk_last_error", "hk_sync"}
    assert need <= seen, need - seen


def test_constructor_calls_pass_one_value_per_field():
    structs = _structs()
    checked = 0
    for m in re.finditer(r"(?<![\w{.])(Hk\w+)\(", SRC):
        name = m.group(1)
        line_start = SRC.rfind("\n", 0, m.start()) + 1
        if name not in structs or SRC[line_start:m.start()].strip().startswith("struct"):
            continue
        args = _call_args(SRC, m.end() - 1)
        assert len(args) == len(structs[name]), (name, len(args), len(structs[name]), SRC[m.start():m.start() + 80])
        checked += 1
    assert checked >= 30


def test_shim_defines_what_integration_md_promises():
    for needle in ("function flatten_scene(ctx::Ptr{Cvoid}, scene)", "function camera_record(cam::Hikari.MatrixCamera)", "function camera_record(cam::Hikari.PerspectiveCamera)",
                   "Hikari.render!(vp::MI355XVolPath", "function Hikari.clear!(vp::MI355XVolPath)", "function Base.close(vp::MI355XVolPath)",
                   "function postprocess!(vp::MI355XVolPath", "function denoise!(vp::MI355XVolPath", "function fill_aux_buffers!(vp::MI355XVolPath",
                   "function material_record(", "function light_record(", "function medium_record(", "function instance_blas("):
        assert needle in SRC, needle
    assert "function flatten_scene end" not in SRC
    # every material / light / medium kind of the header is produced somewhere in the shim
    for const in ("HK_MAT_MATTE", "HK_MAT_MIRROR", "HK_MAT_GLASS", "HK_MAT_CONDUCTOR", "HK_MAT_COATED_DIFFUSE", "HK_MAT_THIN_DIELECTRIC", "HK_MAT_DIFFUSE_TRANSMISSION",
                  "HK_MAT_COATED_DIFFUSE_TRANSMISSION", "HK_MAT_COATED_CONDUCTOR", "HK_MAT_MIX", "HK_MAT_FALLBACK", "HK_LIGHT_POINT", "HK_LIGHT_SPOT", "HK_LIGHT_DIRECTIONAL",
                  "HK_LIGHT_SUN", "HK_LIGHT_AMBIENT", "HK_LIGHT_ENVIRONMENT", "HK_LIGHT_DIFFUSE_AREA", "HK_MEDIUM_HOMOGENEOUS", "HK_MEDIUM_GRID", "HK_MEDIUM_RGB_GRID",
                  "HK_MEDIUM_NANOVDB"):
        assert len(re.findall(r"\b%s\b" % const, SRC)) >= 2, const
    # the enum constants carry the header's values
    hdr = open(os.path.join(ROOT, "include", "hikari_mi355x.h")).read()
    for const, val in re.findall(r"\b(HK_(?:MAT|LIGHT|MEDIUM|SPEC)_\w+)\s*=\s*(\d+)", hdr):
        m = re.search(r"const ([\w, ]*\b%s\b[\w, ]*) = ([^\n]*)" % const, SRC)
        if m:
            names = [n.strip() for n in m.group(1).split(",")]
            vals = re.findall(r"Int32\((\d+)\)", m.group(2))
            assert vals[names.index(const)] == val, const


def test_flatten_scene_fills_every_descriptor_field():
    """hk_scene_desc has 8 counts and 12 array pointers: flatten_scene's HkSceneDesc(...) call must name a Julia array for each of
    them (a count and a pointer per array, in the header's order), and every one of those arrays must be built in the function."""
    i = SRC.index("function flatten_scene(ctx::Ptr{Cvoid}, scene)")
    body = SRC[i:SRC.index("\nend\n", i)]
    j = body.index("HkSceneDesc(")
    args = [a.strip() for a in _call_args(body, j + len("HkSceneDesc"))]
    hdr = open(os.path.join(ROOT, "include", "hikari_mi355x.h")).read()
    fields = re.findall(r"^\s+(?:int32_t|const [\w ]+\*)\s+(\w+);", hdr[hdr.index("typedef struct hk_scene_desc {"):hdr.index("} hk_scene_desc;")], re.M)
    assert len(args) == len(fields) == 20
    counts = {"n_triangles": "meta", "n_materials": "materials", "n_textures": "textures", "n_media_interfaces": "mis", "n_lights": "lights",
              "n_envmaps": "envmaps", "n_media": "media", "n_spectra": "spectra"}
    ptrs = {"positions": "pos", "normals": "nrm", "uvs": "uvs", "tangents": "tan", "meta": "meta", "materials": "materials", "textures": "textures",
            "media_interfaces": "mis", "lights": "lights", "envmaps": "envmaps", "media": "media", "spectra": "spectra"}
    for f, a in zip(fields, args):
        want = "length(%s)" % counts[f] if f in counts else "p(%s)" % ptrs[f]
        assert a == want, (f, a, want)
    for arr in set(ptrs.values()):
        assert re.search(r"\b%s\b\s*(=|,)" % arr, body[:j]), arr      # built (assigned) before the descriptor is


def test_runtests_jl_covers_the_reference_integration_scene():
    rt = open(os.path.join(ROOT, "julia", "test", "runtests.jl")).read()
    for needle in ("Hikari.VolPath(samples=4, max_depth=5)", "HikariMI355X.MI355XVolPath(samples=4, max_depth=5)", "rel_mse(got, ref) <= 1f-3", "frac_within(got, ref) >= 0.99",
                   "Hikari.render!(vp, scene, film, camera)", "PointLight(Point3f(0f0, 1.8f0, 0f0)"):
        assert needle in rt, needle
    # the shim defines what the script calls
    for needle in ("MI355XVolPath(;", "function Base.close(vp::MI355XVolPath)", "function Hikari.clear!(vp::MI355XVolPath)"):
        assert needle in SRC, needle
