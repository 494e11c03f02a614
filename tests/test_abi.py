"""The C-ABI library must load without a GPU and export exactly what include/corona_mi.h declares."""
import ctypes as C
import re

from helpers import REPO, load_pkg

pkg = load_pkg()


def declared_functions():
    text = (REPO / "include" / "corona_mi.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported():
    names = declared_functions()
    assert "mi_render" in names and "mi_scene_create" in names
    lib = C.CDLL(str(pkg.MI_LIB))
    for n in names:
        assert hasattr(lib, n), f"{n} declared in corona_mi.h but not exported"
    assert sorted(pkg.MI_SYMBOLS) == names


def test_struct_sizes_match_header():
    # sizes as laid out by the C compiler for include/corona_mi.h (x86-64)
    assert C.sizeof(pkg.MiNode) == 144
    assert C.sizeof(pkg.MiShadeOp) == 32 and C.sizeof(pkg.MiMaterial) == 160
    assert C.sizeof(pkg.MiCamera) == 128
    assert C.sizeof(pkg.MiPathVertex) == 112 and C.sizeof(pkg.MiPathSplat) == 24
    assert C.sizeof(pkg.MiPathRecord) == 40 + 8 * 24 + 8 * 112
    ray, hit = pkg.ray_dtypes()
    assert ray.itemsize == 32 and hit.itemsize == 32          # mi_ray, mi_hit
    # mi_hero_ext: lambda[4] + six per-vertex quantities [8][4] + splat_value[8][4], floats; = oracle_hero_ext = the dump harness' extension block
    assert pkg.hero_ext_dtype().itemsize == 4 * (4 + 6 * 8 * 4 + 8 * 4) == 912
    import re
    hdr = (REPO / "include" / "corona_mi.h").read_text()
    assert re.search(r"#define MI_WAVELENGTHS_HERO 4\b", hdr) and "mi_scene_set_wavelengths" in hdr and "mi_trace_paths_hero" in hdr


def test_product_never_imports_oracle():
    """The product path must not route through the CPU oracle."""
    for p in (REPO / "corona-13_amd").rglob("*"):
        if p.suffix in (".py", ".c", ".h", ".hip", ".cpp") or p.name == "Makefile":
            txt = p.read_text(errors="ignore")
            assert "liboracle" not in txt and "oracle/" not in txt.replace("oracle/_ref", ""), p


def test_plan_of_cfg5_on_eight_gpus():
    """mi_plan_launches (no device touched): the launches mi_group_render queues -- the group's share arithmetic and the per-member
    launch splitting that keeps a workgroup's part below 2^31 path indices -- for configs[4]: 3840x2160 (padded 3840x2176), 1024 spp,
    8 x MI355X with 256 resident workgroups each; plus the corners: remainders, a range smaller than the group, a range above
    256 * 2^31 on one device."""
    lib = C.CDLL(str(pkg.MI_LIB))

    class Launch(C.Structure):
        _fields_ = [("member", C.c_int32), ("grid", C.c_int32), ("first", C.c_uint64), ("count", C.c_uint64)]

    lib.mi_plan_launches.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.POINTER(Launch), C.c_int]
    lib.mi_plan_launches.restype = C.c_int

    def plan(first, count, members, grid):
        buf = (Launch * 64)()
        n = lib.mi_plan_launches(first, count, members, grid, buf, 64)
        assert 0 <= n <= 64
        return [(b.member, b.grid, b.first, b.count) for b in buf[:n]]

    def check(first, count, members, grid):
        p = plan(first, count, members, grid)
        # the launches tile [first, first + count) in order, member by member, with nothing lost and nothing twice
        pos = first
        for member, g, f, c in p:
            assert f == pos and c > 0 and 1 <= g <= grid
            assert c <= (g << 31) and -(-c // g) <= (1 << 31)                # every workgroup's part fits its 32-bit counter (which runs a little past it)
            assert g == min(grid, -(-c // 1024))                             # a small range starts fewer workgroups
            pos += c
        assert pos == first + count
        shares = [sum(c for m, _, _, c in p if m == k) for k in range(members)]
        assert max(shares) - min(shares) <= 1 and sorted(shares, reverse=True) == shares      # remainder to the lowest members
        assert [m for m, _, _, _ in p] == sorted(m for m, _, _, _ in p)
        return p

    job = 1024 * 3840 * 2176
    p = check(0, job, 8, 256)
    assert len(p) == 8 and all(c == job // 8 for _, _, _, c in p)           # 1.07 G paths per GPU: one launch each (256 * 2^31 = 550 G)
    check(12345, job + 5, 8, 256)
    check(7, 5, 8, 256)                                                      # fewer paths than members: three members get nothing
    assert len(plan(7, 5, 8, 256)) == 5
    big = (256 << 31) * 2 + 123                                              # one device, more than two full launches
    p = check(0, big, 1, 256)
    assert [c for _, _, _, c in p] == [256 << 31, 256 << 31, 123] and p[-1][1] == 1
    check(0, (256 << 31) * 3 + 1000, 2, 256)
    assert lib.mi_plan_launches(0, 10, 0, 256, None, 0) < 0                   # bad argument
