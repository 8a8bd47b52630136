"""Host-side mirror of Hikari.Scene (src/scene.jl:17-161, src/scene-mesh.jl:9-179): incremental `push`, then
`sync` flattens everything into the POD arrays of hk_scene_desc (the job the Julia shim does by walking
`scene.accel` / `scene.materials` / `scene.lights`)."""
import ctypes as C

import numpy as np

from . import _abi as A
from . import geometry as G
from . import lights as L
from . import materials as M

f32 = np.float32


def _luminance(c):
    return float(f32(0.212671) * f32(c[0]) + f32(0.715160) * f32(c[1]) + f32(0.072169) * f32(c[2]))


class Scene:
    def __init__(self):
        self.lights = []            # in push order; flattened by type slot in sync()
        self._light_types = []      # type-slot order (MultiTypeSet: first-seen type order)
        self.materials = []         # BSDF materials in push order
        self._material_types = []
        self._material_keys = []    # SetKey (type_idx, vec_idx) per material, 1-based like the reference
        self.media = []
        self.media_interfaces = []  # (material, inside, outside) 0-based, -1 = vacuum
        self._meshes = []           # (Mesh world-space, per-face metas)
        self.textures = []
        self.spectra = []
        self._envmaps = []
        self._desc = None
        self._keep = None
        self._device = {}           # id(ctx) -> hk_scene handle (owned: released by close() / the next sync())
        self.bounds = None

    # ---- push! ---------------------------------------------------------------------------------
    def push_light(self, light):
        if type(light) not in self._light_types:
            self._light_types.append(type(light))
        self.lights.append(light)
        return len(self.lights)

    def _push_material_record(self, mat):
        if type(mat) not in self._material_types:
            self._material_types.append(type(mat))
        t = self._material_types.index(type(mat)) + 1
        v = sum(1 for m in self.materials if type(m) is type(mat)) + 1
        if isinstance(mat, M.MixMaterial):
            mat._idx1 = self._push_material_record(mat.material1)
            mat._idx2 = self._push_material_record(mat.material2)
            v = sum(1 for m in self.materials if type(m) is type(mat)) + 1
        self.materials.append(mat)
        self._material_keys.append((t, v))
        return len(self.materials) - 1

    def _push_medium(self, medium):
        if medium is None:
            return -1
        for i, m in enumerate(self.media):
            if m is medium:
                return i
        self.media.append(medium)
        return len(self.media) - 1

    def push_material(self, material):
        """push!(scene, material) -> index into media_interfaces (scene.jl:80-100)."""
        mi = material if isinstance(material, M.MediumInterface) else M.MediumInterface(material)
        mat_idx = self._push_material_record(mi.material)
        triple = (mat_idx, self._push_medium(mi.inside), self._push_medium(mi.outside))
        if triple not in self.media_interfaces:
            self.media_interfaces.append(triple)
        return self.media_interfaces.index(triple)

    def push(self, obj, material=None, transform=None):
        if isinstance(obj, L.Light):
            return self.push_light(obj)
        mesh = obj
        mi_idx = self.push_material(material)
        n = mesh.n_faces
        metas = np.zeros((n, 3), dtype=np.uint32)
        metas[:, 0] = mi_idx
        metas[:, 1] = np.arange(1, n + 1)
        emission = self._emission_info(material)
        if emission is not None:
            self._register_face_area_lights(mesh, metas, emission)
        world = mesh if transform is None else mesh.transformed(transform)
        self._meshes.append((world, metas))
        self._desc = None
        return mi_idx

    @staticmethod
    def _emission_info(material):
        if isinstance(material, M.MediumInterface):
            if material.emission is not None:
                return material.emission
            return Scene._emission_info(material.material)
        if isinstance(material, M.Emissive):
            return material
        return None

    def _register_face_area_lights(self, mesh, metas, em):
        """scene-mesh.jl:98-131 — uses the UN-transformed mesh vertices (quirk Q18)."""
        for i in range(mesh.n_faces):
            vs = mesh.positions[i]
            uv = mesh.uvs[i] if mesh.uvs is not None else np.array([[0, 0], [1, 0], [1, 1]], dtype=f32)
            Le = em.Le
            if isinstance(Le, M.Texture):
                # evaluate_face_emission (scene-mesh.jl:49): the texture is point-sampled ONCE at the face's centroid uv
                # (evaluate_texture -> _sample_texture_data: nearest texel, (1-v, u) flip); the light carries that constant
                cu = f32(f32(f32(uv[0][0] + uv[1][0]) + uv[2][0]) / f32(3))
                cv = f32(f32(f32(uv[0][1] + uv[1][1]) + uv[2][1]) / f32(3))
                d = Le.data
                th, tw = d.shape[:2]
                ti = min(max(int(f32(1) + f32(th - 1) * f32(f32(1) - cv)), 1), th)
                tj = min(max(int(f32(1) + f32(tw - 1) * cu), 1), tw)
                texel = d[ti - 1, tj - 1]
                Le = M.RGBSpectrum(*[float(x) for x in np.atleast_1d(texel)[:4]]) if np.ndim(texel) else M.RGBSpectrum(float(texel))
            if _luminance(Le.c) < 1e-4:
                continue
            e1, e2 = (vs[1] - vs[0]).astype(f32), (vs[2] - vs[0]).astype(f32)
            cp = np.array([e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]], dtype=f32)
            twice = np.sqrt(f32(f32(cp[0] * cp[0]) + f32(cp[1] * cp[1])) + f32(cp[2] * cp[2]), dtype=f32)
            if twice < 1e-10:
                continue
            normal = (cp / twice).astype(f32)
            area = f32(0.5) * twice
            self.push_light(L.DiffuseAreaLight(vs.copy(), normal, float(area), uv.copy(), Le, em.scale, em.two_sided))
            metas[i, 2] = len(self.lights)  # flat index = length(scene.lights) at push time

    # ---- sync! / flatten -------------------------------------------------------------------------
    def flat_lights(self):
        """flat_to_light_index order: type slots in first-seen order (lights/light-sampler.jl:289-329)."""
        out = []
        for t in self._light_types:
            out.extend(l for l in self.lights if type(l) is t)
        return out

    def _tex_rgba(self, v, keep):
        r = A.hk_tex_rgba()
        if isinstance(v, M.Texture):
            r.tex = self._texture_index(v, keep)
            r.c[:] = (0, 0, 0, 1)
        else:
            r.tex = -1
            r.c[:] = v.c
        return r

    def _tex_f32(self, v, keep):
        r = A.hk_tex_f32()
        if isinstance(v, M.Texture):
            r.tex = self._texture_index(v, keep)
            r.v = 0.0
        else:
            r.tex = -1
            r.v = float(f32(v))
        return r

    def _texture_index(self, tex, keep):
        for i, t in enumerate(self.textures):
            if t is tex:
                return i
        self.textures.append(tex)
        return len(self.textures) - 1

    def _envmap_index(self, env):
        for i, e in enumerate(self._envmaps):
            if e is env:
                return i
        self._envmaps.append(env)
        return len(self._envmaps) - 1

    def _spectrum_index(self, sp):
        for i, s in enumerate(self.spectra):
            if s is sp:
                return i
        self.spectra.append(sp)
        return len(self.spectra) - 1

    def _material_record(self, m, key, keep):
        r = A.hk_material()
        r.kind = m.kind
        r.flags = 0
        for k in range(4):
            r.rgb[k].tex = -1
            r.rgb[k].c[:] = (0, 0, 0, 1)
        for k in range(8):
            r.f[k].tex = -1
        r.spectrum[0] = r.spectrum[1] = -1
        T, F = self._tex_rgba, self._tex_f32
        if isinstance(m, M.MatteMaterial):
            r.rgb[0], r.f[0] = T(m.Kd, keep), F(m.sigma, keep)
        elif isinstance(m, M.MirrorMaterial):
            r.rgb[0] = T(m.Kr, keep)
        elif isinstance(m, M.GlassMaterial):
            r.rgb[0], r.rgb[1], r.f[0] = T(m.Kr, keep), T(m.Kt, keep), F(m.index, keep)
        elif isinstance(m, M.ConductorMaterial):
            for slot, v in ((0, m.eta), (1, m.k)):
                if isinstance(v, M.PiecewiseLinearSpectrum):
                    r.spectrum[slot] = self._spectrum_index(v)
                else:
                    r.rgb[slot] = T(v, keep)
            r.f[0] = F(m.roughness, keep)
            r.flags = A.HK_MATF_REMAP_ROUGHNESS if m.remap_roughness else 0
        elif isinstance(m, M.CoatedDiffuseMaterial):
            r.rgb[0], r.rgb[1] = T(m.reflectance, keep), T(m.albedo, keep)
            for k, v in enumerate((m.u_roughness, m.v_roughness, m.thickness, m.eta, m.g)):
                r.f[k] = F(v, keep)
            r.i[0], r.i[1] = m.max_depth, m.n_samples
            r.flags = A.HK_MATF_REMAP_ROUGHNESS if m.remap_roughness else 0
        elif isinstance(m, M.ThinDielectricMaterial):
            r.f[0] = F(m.eta, keep)
        elif isinstance(m, M.DiffuseTransmissionMaterial):
            r.rgb[0], r.rgb[1], r.f[0] = T(m.reflectance, keep), T(m.transmittance, keep), F(m.scale, keep)
        elif isinstance(m, M.CoatedDiffuseTransmissionMaterial):
            r.rgb[0], r.rgb[1], r.rgb[2] = T(m.reflectance, keep), T(m.transmittance, keep), T(m.albedo, keep)
            for k, v in enumerate((m.u_roughness, m.v_roughness, m.thickness, m.eta, m.g)):
                r.f[k] = F(v, keep)
            r.i[0], r.i[1] = m.max_depth, m.n_samples
            r.flags = A.HK_MATF_REMAP_ROUGHNESS if m.remap_roughness else 0
        elif isinstance(m, M.CoatedConductorMaterial):
            for slot, v in ((0, m.conductor_eta), (1, m.conductor_k)):
                if isinstance(v, M.PiecewiseLinearSpectrum):
                    r.spectrum[slot] = self._spectrum_index(v)
                else:
                    r.rgb[slot] = T(M._rgb(v), keep)
            r.rgb[2], r.rgb[3] = T(m.reflectance, keep), T(m.albedo, keep)
            for k, v in enumerate((m.interface_u_roughness, m.interface_v_roughness, m.interface_eta,
                                   m.conductor_u_roughness, m.conductor_v_roughness, m.thickness, m.g)):
                r.f[k] = F(v, keep)
            r.i[0], r.i[1] = m.max_depth, m.n_samples
            r.flags = (A.HK_MATF_REMAP_ROUGHNESS if m.remap_roughness else 0) | (A.HK_MATF_USE_ETA_K if m.use_eta_k else 0)
        elif isinstance(m, M.MixMaterial):
            r.f[0] = F(m.amount, keep)
            r.i[0], r.i[1] = m._idx1, m._idx2
            k1, k2 = self._material_keys[m._idx1], self._material_keys[m._idx2]
            r.mix_key[:] = (k1[0], k1[1], k2[0], k2[1])
        return r

    def _light_record(self, l, keep):
        r = A.hk_light()
        r.kind = l.kind
        r.envmap = -1
        r.Le.tex = -1
        if isinstance(l, L.DiffuseAreaLight):
            r.spectrum_kind = A.HK_SPEC_RGB
            r.scale = l.scale
            r.v[:] = [float(x) for x in np.asarray(l.vertices, dtype=f32).reshape(-1)]
            r.normal[:] = [float(x) for x in l.normal]
            r.area = l.area
            r.uv[:] = [float(x) for x in np.asarray(l.uv, dtype=f32).reshape(-1)]
            r.Le = self._tex_rgba(l.Le, keep)
            r.two_sided = 1 if l.two_sided else 0
            return r
        if l.kind == A.HK_LIGHT_ENVIRONMENT:
            r.spectrum_kind = A.HK_SPEC_RGB
            r.i_rgb[:] = l.scale_rgb.c
            r.scale = 1.0
            r.envmap = self._envmap_index(l.env_map)
            return r
        sf = L._spec_fields(l.i)
        r.spectrum_kind = sf["spectrum_kind"]
        r.i_rgb[:] = sf["i_rgb"]
        r.poly[:] = sf["poly"]
        r.illum_scale = sf["illum_scale"]
        r.scale = l.scale
        if hasattr(l, "position"):
            r.position[:] = l.position
        if hasattr(l, "direction"):
            r.direction[:] = l.direction
        if isinstance(l, L.SpotLight):
            r.world_to_light[:] = [float(x) for x in l.world_to_light.reshape(-1)]
            r.light_to_world[:] = [float(x) for x in l.light_to_world.reshape(-1)]
            r.cos_total_width, r.cos_falloff_start = l.cos_total_width, l.cos_falloff_start
        return r

    def sync(self):
        """sync!(scene): flatten to hk_scene_desc (kept alive on self) and compute world bounds."""
        self.close()                # device scenes built from the previous description are released, not leaked
        keep = []
        P = np.concatenate([m.positions for m, _ in self._meshes], axis=0) if self._meshes else np.zeros((0, 3, 3), f32)
        T = P.shape[0]
        any_n = any(m.normals is not None for m, _ in self._meshes)
        any_uv = any(m.uvs is not None for m, _ in self._meshes)
        Nn = Uv = None
        if any_n:
            Nn = np.concatenate([m.normals if m.normals is not None else np.full(m.positions.shape, np.nan, f32) for m, _ in self._meshes], axis=0)
        if any_uv:
            default_uv = np.array([[0, 0], [1, 0], [1, 1]], dtype=f32)
            Uv = np.concatenate([m.uvs if m.uvs is not None else np.broadcast_to(default_uv, (m.n_faces, 3, 2)) for m, _ in self._meshes], axis=0)
        metas = np.concatenate([mt for _, mt in self._meshes], axis=0) if self._meshes else np.zeros((0, 3), np.uint32)
        P = np.ascontiguousarray(P, dtype=f32)
        metas = np.ascontiguousarray(metas, dtype=np.uint32)
        mats = (A.hk_material * max(1, len(self.materials)))()
        for i, m in enumerate(self.materials):
            mats[i] = self._material_record(m, self._material_keys[i], keep)
        mis = (A.hk_medium_interface * max(1, len(self.media_interfaces)))()
        for i, (a, b, c) in enumerate(self.media_interfaces):
            mis[i].material, mis[i].inside, mis[i].outside = a, b, c
        fl = self.flat_lights()
        self._envmaps = []
        lts = (A.hk_light * max(1, len(fl)))()
        for i, l in enumerate(fl):
            lts[i] = self._light_record(l, keep)
        texs = (A.hk_texture * max(1, len(self.textures)))()
        for i, t in enumerate(self.textures):
            d = t.data
            if isinstance(t, M.VertexColorTexture):      # Julia face_colors[3, n_faces]: a face's 3 colours are adjacent
                keep.append(d)
                texs[i].width, texs[i].height, texs[i].channels, texs[i].kind = t.n_faces, 3, 4, 1
                texs[i].data = d.ctypes.data_as(A.PF)
                continue
            h, w = d.shape[:2]
            ch = 1 if d.ndim == 2 else d.shape[2]
            jl = np.ascontiguousarray(np.transpose(d.reshape(h, w, ch), (1, 0, 2)))  # [w][h][c] == Julia [h,w] column-major
            keep.append(jl)
            texs[i].width, texs[i].height, texs[i].channels = w, h, ch
            texs[i].data = jl.ctypes.data_as(A.PF)
        specs = (A.hk_pl_spectrum * max(1, len(self.spectra)))()
        for i, s in enumerate(self.spectra):
            specs[i].n = s.lambdas.size
            specs[i].lambdas = s.lambdas.ctypes.data_as(A.PF)
            specs[i].values = s.values.ctypes.data_as(A.PF)
        media_recs, media_keep = _media_records(self.media)
        keep.append(media_keep)
        d = A.hk_scene_desc()
        d.n_triangles, d.n_materials, d.n_textures = T, len(self.materials), len(self.textures)
        envs = (A.hk_envmap * max(1, len(self._envmaps)))()
        for i, e in enumerate(self._envmaps):
            envs[i] = e.record()
        d.n_media_interfaces, d.n_lights, d.n_envmaps, d.n_media, d.n_spectra = len(self.media_interfaces), len(fl), len(self._envmaps), len(self.media), len(self.spectra)
        d.positions = P.ctypes.data_as(A.PF)
        d.normals = Nn.ctypes.data_as(A.PF) if Nn is not None else None
        d.uvs = np.ascontiguousarray(Uv, dtype=f32).ctypes.data_as(A.PF) if Uv is not None else None
        if Uv is not None:
            Uv = np.ascontiguousarray(Uv, dtype=f32)
            d.uvs = Uv.ctypes.data_as(A.PF)
        if Nn is not None:
            Nn = np.ascontiguousarray(Nn, dtype=f32)
            d.normals = Nn.ctypes.data_as(A.PF)
        d.tangents = None
        d.meta = metas.ctypes.data_as(C.POINTER(A.hk_tri_meta))
        d.materials, d.textures, d.media_interfaces, d.lights = mats, texs, mis, lts
        d.envmaps, d.media, d.spectra = envs, media_recs, specs
        self._keep = (P, Nn, Uv, metas, mats, mis, lts, texs, specs, keep, media_recs, envs, list(self._envmaps))
        self._desc = d
        if T:
            lo, hi = P.reshape(-1, 3).min(axis=0), P.reshape(-1, 3).max(axis=0)
            c = (lo + hi) * f32(0.5)
            self.bounds = (lo, hi, c, float(np.linalg.norm(hi - c)))
        self._device = {}
        return self

    def close(self):
        """Release every device scene (BVH, leaf triangles, textures, env maps, media) created from this Scene."""
        dev, self._device = getattr(self, "_device", None) or {}, {}
        if dev:
            from . import _lib
            L = _lib.lib()
            for h in dev.values():
                L.hk_scene_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def desc(self):
        if self._desc is None:
            self.sync()
        return self._desc

    def world_radius(self):
        self.desc
        return self.bounds[3]


def _media_records(media):
    recs = (A.hk_medium * max(1, len(media)))()
    keep = []
    for i, m in enumerate(media):
        m.fill_record(recs[i], keep)
    return recs, keep
