"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every
symbol include/splat_hip.h declares, the ctypes structs mirror the header, and the
Python surface has the reference's shape and error behaviour.  No kernel launches."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "splat_hip.h")).read()


def _declared_functions():
    body = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    return sorted(set(re.findall(r"\b(splat_[a-z_0-9]+)\s*\(", body)))


def test_library_exports_every_declared_symbol():
    from splatam_amd import _capi
    L = _capi.lib()
    names = _declared_functions()
    assert len(names) >= 12
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/splat_hip.h but not exported"
    assert set(_capi.EXPORTS) <= set(names)
    assert L.splat_abi_version() == int(re.search(r"#define SPLAT_ABI_VERSION (\d+)", HEADER).group(1))
    assert L.splat_error_string(0) == b"ok" and L.splat_error_string(1) == b"invalid argument"
    assert L.splat_num_tiles(1200, 680) == 75 * 43 and L.splat_num_tiles(0, 10) == 0


def _struct_fields(name):
    m = re.search(r"typedef struct " + name + r" \{(.*?)\} " + name + ";", HEADER, flags=re.S)
    body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = re.sub(r"\[[^\]]*\]", "", decl)                  # array extents
        if decl.strip():
            names += [re.findall(r"[A-Za-z_0-9]+", part)[-1] for part in decl.split(",")]
    return names


@pytest.mark.parametrize("name", ["SplatCamera", "SplatGaussians", "SplatState", "SplatGrads", "SplatMap", "SplatFrameData",
                                  "SplatLossConfig", "SplatIterWorkspace", "SplatAdamMap", "SplatMapStore", "SplatAddArgs",
                                  "SplatPruneArgs", "SplatPoseAdam", "SplatDensifyArgs"])
def test_ctypes_structs_mirror_header(name):
    from splatam_amd import _capi
    assert [f[0] for f in getattr(_capi, name)._fields_] == _struct_fields(name)
    # ... and the compiled layout (field types, padding): sizeof as the library sees it
    assert _capi.lib().splat_sizeof(name.encode()) == C.sizeof(getattr(_capi, name)) > 0
    assert getattr(_capi, name) in _capi.MIRRORED_STRUCTS


def test_sizeof_of_an_unknown_struct_is_zero():
    from splatam_amd import _capi
    assert _capi.lib().splat_sizeof(b"NoSuchStruct") == 0 and _capi.lib().splat_sizeof(None) == 0


def test_constants_match_header():
    from splatam_amd import _capi
    for k in ("SPLAT_TILE", "SPLAT_MAX_CHANNELS", "SPLAT_GRAD_STRIDE", "SPLAT_COUNTER_STRIDE"):
        assert getattr(_capi, k) == int(re.search(rf"#define {k} (\d+)", HEADER).group(1))


def test_invalid_arguments_return_codes_not_crashes():
    from splatam_amd import _capi
    L = _capi.lib()
    cam, g, st = _capi.SplatCamera(), _capi.SplatGaussians(), _capi.SplatState()
    assert L.splat_preprocess_forward(C.byref(cam), C.byref(g), C.byref(st), None) == 1      # zero-size image
    cam.image_width, cam.image_height = 64, 64
    g.P, g.channels = 10, 3
    assert L.splat_preprocess_forward(C.byref(cam), C.byref(g), C.byref(st), None) == 1      # null matrices
    assert L.splat_mark_visible(-1, None, None, None, None) == 1
    g.channels = 99
    assert L.splat_render_forward(C.byref(cam), C.byref(g), C.byref(st), None, None, None) == 1
    # bucketed lists behind the reference API exist only as group binning: a state that asks for buckets without the group arrays (or
    # with a list-length hint beyond what the composite sorts) is refused before anything is launched
    lay = _capi.state_layout(10, 64, 64, 1, 16 * 1024, _capi.SPLAT_LAYOUT_GROUPS)
    st2 = _capi.SplatState()
    assert L.splat_state_bind(C.byref(st2), None, 1 << 20, lay.arrays, lay.n, 1, 16 * 1024) == 0
    cam.viewmatrix = cam.projmatrix = 1 << 12
    g.channels = 3
    g.means3D = g.opacities = g.colors_precomp = g.scales = g.rotations = 1 << 12
    st2.tile_stride, st2.group_stride, st2.max_list_hint = 1024, 0, 100                     # no group records
    assert L.splat_preprocess_forward(C.byref(cam), C.byref(g), C.byref(st2), None) == 1
    st2.group_stride, st2.max_list_hint = 4096, 2000                                        # lists the composite cannot sort
    assert L.splat_preprocess_forward(C.byref(cam), C.byref(g), C.byref(st2), None) == 1
    st2.max_list_hint, st2.tile_stride = 100, 4096                                          # buckets beyond the capacity
    assert L.splat_preprocess_forward(C.byref(cam), C.byref(g), C.byref(st2), None) == 1


def test_python_surface_matches_reference_shape():
    from diff_gaussian_rasterization import GaussianRasterizationSettings as Camera
    from diff_gaussian_rasterization import GaussianRasterizer as Renderer
    assert Camera._fields == ('image_height', 'image_width', 'tanfovx', 'tanfovy', 'bg', 'scale_modifier', 'viewmatrix',
                              'projmatrix', 'sh_degree', 'campos', 'prefiltered')
    cam = Camera(image_height=8, image_width=8, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(3), scale_modifier=1.0,
                 viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0, campos=torch.zeros(3), prefiltered=False)
    r = Renderer(raster_settings=cam)
    assert isinstance(r, torch.nn.Module) and r.raster_settings is cam
    z = torch.zeros(2, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(means3D=z, means2D=z, opacities=torch.ones(2, 1), scales=z, rotations=torch.ones(2, 4))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
        r(means3D=z, means2D=z, opacities=torch.ones(2, 1), colors_precomp=z)
    # no silent CPU path: CPU tensors fail loudly
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(means3D=z, means2D=z, opacities=torch.ones(2, 1), colors_precomp=z, scales=z, rotations=torch.ones(2, 4))


def test_product_does_not_import_the_oracle():
    for pkg in ("splatam_amd", "diff_gaussian_rasterization"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, pkg)):
            for f in files:
                if f.endswith((".py", ".hip", ".h")):
                    src = open(os.path.join(dirpath, f)).read()
                    assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                    assert "raster_ref" not in src, f


def test_scratch_layouts_describe_the_structs():
    """include/splat_hip.h "Scratch layouts": a binding sizes and wires the scratch without any Python arithmetic.  Every array a
    SplatState / SplatIterWorkspace points at is named; offsets are SPLAT_SLAB_ALIGN-aligned, disjoint and inside the slab;
    splat_workspace_bytes is the layout's total; bind() writes slab + offset into exactly those fields (no GPU needed: it only forms
    pointers)."""
    import ctypes as C
    from splatam_amd import _capi
    L = _capi.lib()
    P, W, H, cap = 12345, 1200, 680, 700_000
    T = L.splat_num_tiles(W, H)
    lay = _capi.state_layout(P, W, H, 1, cap, _capi.SPLAT_LAYOUT_LONG_LISTS | _capi.SPLAT_LAYOUT_BACKWARD)
    assert lay.total == L.splat_workspace_bytes(P, W, H, cap) and lay.total % _capi.SPLAT_SLAB_ALIGN == 0
    want = {"depth": 4 * P, "xy": 8 * P, "conic_opacity": 16 * P, "rect": 8 * P, "radii": 4 * P, "tile_base": 4 * (T + 1),
            "tile_count": 4 * T * _capi.SPLAT_COUNTER_STRIDE, "tile_cursor": 4 * T * _capi.SPLAT_COUNTER_STRIDE, "keys": 8 * cap,
            "point_list": 4 * cap, "keys_alt": 8 * cap, "long_items": 4 * (cap // 1024 + T + 1), "long_base": 4 * (T + 1),
            "final_T": 4 * W * H, "n_contrib": 4 * W * H, "status": 16, "accum": 4 * _capi.SPLAT_GRAD_STRIDE * P}
    assert lay.bytes == want
    spans = sorted((lay.offset[k], lay.offset[k] + lay.bytes[k]) for k in lay.names)
    assert all(o % _capi.SPLAT_SLAB_ALIGN == 0 for o, _ in spans) and spans[-1][1] <= lay.total
    assert all(spans[i][1] <= spans[i + 1][0] for i in range(len(spans) - 1))
    assert "rgb" not in lay.bytes and "rgb" in _capi.state_layout(P, W, H, 1, cap, _capi.SPLAT_LAYOUT_SH).bytes
    assert _capi.state_layout(P, W, H, 16, cap, 0).bytes["tile_count"] == 16 * want["tile_count"]
    st, gr = _capi.SplatState(), _capi.SplatGrads()
    base = 1 << 20                                               # any aligned address: bind only forms pointers
    assert L.splat_state_bind(C.byref(st), C.byref(gr), base, lay.arrays, lay.n, 1, cap) == 0
    for k in lay.names:
        got = gr.accum if k == "accum" else getattr(st, k)
        assert got == base + lay.offset[k], k
    assert st.capacity == cap and st.sub_bins == 1 and st.group_recs is None
    assert L.splat_state_bind(C.byref(st), None, base + 8, lay.arrays, lay.n, 1, cap) != 0       # a misaligned slab is refused
    assert L.splat_state_layout(-1, W, H, 1, cap, 0, None, 0, None) < 0 and L.splat_state_layout(P, W, H, 3, cap, 0, None, 0, None) < 0
    # group binning behind the reference API (ABI 10): no key buckets, the group counters padded to the slab alignment with the status
    # words exactly behind them (the library zeroes both with one memset), records and accumulator on request
    Wg, Hg = 1232, 720                                           # 77 x 45 tiles -> 39 x 23 = 897 groups: an odd multiple of 128 bytes
    Tg = L.splat_num_tiles(Wg, Hg)
    gl = _capi.state_layout(P, Wg, Hg, 1, Tg * 1024, _capi.SPLAT_LAYOUT_GROUPS | _capi.SPLAT_LAYOUT_RECS | _capi.SPLAT_LAYOUT_BACKWARD)
    Gg = ((Wg + 15) // 16 + 1) // 2 * (((Hg + 15) // 16 + 1) // 2)
    assert "keys" not in gl.bytes and gl.bytes["point_list"] == 4 * Tg * 1024 and gl.bytes["tile_recs"] == 48 * Tg * 1024
    assert gl.bytes["group_recs"] == 16 * Gg * 4 * 1024 and gl.bytes["accum"] == 4 * _capi.SPLAT_GRAD_STRIDE * P
    assert gl.bytes["group_count"] == -(-(4 * Gg * _capi.SPLAT_COUNTER_STRIDE) // 256) * 256 > 4 * Gg * _capi.SPLAT_COUNTER_STRIDE
    assert gl.offset["status"] == gl.offset["group_count"] + gl.bytes["group_count"]
    assert L.splat_state_layout(P, Wg, Hg, 1, 0, _capi.SPLAT_LAYOUT_GROUPS, None, 0, None) < 0       # (needs the bucket capacity: tiles x stride)
    # the fused iteration's workspace
    gs = 4 * 448
    fl = _capi.SPLAT_LAYOUT_SSIM | _capi.SPLAT_LAYOUT_OUTLIER | _capi.SPLAT_LAYOUT_TILE_ORDER
    il = _capi.iter_workspace_layout(P, W, H, cap, gs, fl)
    assert il.total == L.splat_iter_workspace_bytes(P, W, H, cap, gs, fl)
    G = ((W + 15) // 16 + 1) // 2 * (((H + 15) // 16 + 1) // 2)
    assert il.bytes["feat8"] == 32 * P and il.bytes["out6"] == il.bytes["dL_dout6"] == 24 * W * H and il.bytes["ssim_maps"] == 36 * W * H
    assert il.bytes["sums"] == 8 * _capi.SPLAT_ITER_SUM_COPIES * _capi.SPLAT_ITER_SUMS and il.bytes["d_cam"] == 4 * _capi.SPLAT_ITER_DCAM
    assert il.bytes["st.group_count"] == 4 * G * _capi.SPLAT_COUNTER_STRIDE and il.bytes["st.group_recs"] == 16 * G * gs
    assert il.bytes["st.tile_order"] == 4 * 8 * ((T + 7) // 8) and il.bytes["outlier_scratch"] == 4 * L.splat_map_scratch_words(W * H)
    zero = {k for k in il.names if il.zero_init[k]}
    # (the launch order is zero-initialised: a zeroed buffer IS the natural order -- a binding that merely zeroes what the layout marks
    #  can run the iteration)
    assert zero == {"st.radii", "st.tile_count", "st.long_base", "st.group_count", "st.tile_work", "st.tile_order", "st.status", "dL_dout6",
                    "accum", "sums", "d_cam", "outlier_scratch"}
    # ... and it is only laid out (and bound) on request
    plain = _capi.iter_workspace_layout(P, W, H, cap, gs, _capi.SPLAT_LAYOUT_SSIM)
    assert "st.tile_order" not in plain.bytes and "st.tile_work" not in plain.bytes
    ws0 = _capi.SplatIterWorkspace()
    assert L.splat_iter_workspace_bind(C.byref(ws0), base, plain.arrays, plain.n, cap, gs) == 0
    assert ws0.st.tile_order is None and ws0.st.tile_work is None
    ws = _capi.SplatIterWorkspace()
    assert L.splat_iter_workspace_bind(C.byref(ws), base, il.arrays, il.n, cap, gs) == 0
    for k in il.names:
        got = getattr(ws.st, k[3:]) if k.startswith("st.") else getattr(ws, k)
        assert got == base + il.offset[k], k
    assert ws.st.group_stride == gs and ws.st.capacity == cap and ws.d_means3D is None
    assert "ssim_maps" not in _capi.iter_workspace_layout(P, W, H, cap, 0, 0).bytes
