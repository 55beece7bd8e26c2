"""CPU: the Rust `extern "C"` block and #[repr(C)] structs in INTEGRATION.md (source only: there is no rustc in the
image, so the text cannot be compiled) are kept honest mechanically: every function of include/gpe.h appears in the
block with the same number of arguments and a matching scalar / pointer shape per argument, every struct has the
header's fields in the header's order with matching widths, and the three host mirrors (Rust text, C++
gpe_host.hpp, Python engine.py) expose the reference's ParticleSystem policy methods
(/root/reference/src/particles/particle_system.rs:221-247)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "gpe.h")).read()
DOC = open(os.path.join(ROOT, "INTEGRATION.md")).read()


def _strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def _c_functions():
    """name -> list of C parameter declarations (strings), from include/gpe.h"""
    text = _strip_c_comments(HEADER)
    out = {}
    for m in re.finditer(r"\b(?:gpe_status|uint32_t|float|const char \*)\s*\**\s*(gpe_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        name, args = m.group(1), " ".join(m.group(2).split())
        if name.endswith("_fn"):
            continue
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        out[name] = params
    return out


def _rust_functions():
    block = re.search(r'extern "C" \{(.*?)\n\}', DOC, flags=re.S).group(1)
    block = re.sub(r"//[^\n]*", " ", block)
    out = {}
    for m in re.finditer(r"pub fn (gpe_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        args = " ".join(m.group(2).split())
        params = [a.strip() for a in args.split(",") if a.strip()]
        out[m.group(1)] = params
    return out


def _c_shape(decl):
    """'ptr' | 'f32' | 'u64' | 'u32' | 'i32' | 'enum' of one C parameter declaration"""
    d = decl.replace("const ", "")
    if "*" in d or "_fn" in d:
        return "ptr"
    for c, s in (("uint64_t", "u64"), ("uint32_t", "u32"), ("int32_t", "i32"), ("float", "f32"), ("gpe_array", "enum")):
        if d.startswith(c):
            return s
    raise AssertionError("unknown C parameter type: %r" % decl)


def _rust_shape(decl):
    t = decl.split(":", 1)[1].strip()
    if t.startswith("*") or t.startswith("Option<"):
        return "ptr"
    return {"u64": "u64", "u32": "u32", "i32": "i32", "f32": "f32", "gpe_array": "enum"}[t]


def test_extern_block_declares_every_header_function_with_the_same_arguments():
    c, r = _c_functions(), _rust_functions()
    assert len(c) >= 60
    assert sorted(c) == sorted(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name in c:
        assert len(c[name]) == len(r[name]), (name, c[name], r[name])
        assert [_c_shape(a) for a in c[name]] == [_rust_shape(a) for a in r[name]], (name, c[name], r[name])


def _c_struct_fields(name):
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), _strip_c_comments(HEADER), flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"(const )?([a-z0-9_]+)\s+(.+)$", decl)
        assert m, decl
        ctype = m.group(2)
        for item in m.group(3).split(","):                     # `uint32_t rank, world_size, n_slots;` declares three
            im = re.match(r"\s*(\*?)\s*([a-z0-9_]+)(\[(\d+)\])?\s*$", item)
            assert im, decl
            ptr, fname, count = im.group(1), im.group(2), int(im.group(4) or 1)
            width = 8 if ptr else {"uint32_t": 4, "int32_t": 4, "float": 4, "uint64_t": 8, "double": 8, "char": 1, "uint8_t": 1}[ctype]
            fields.append((fname, width, count))
    return fields


def _rust_struct_fields(name):
    body = re.search(r"pub struct %s \{(.*?)\}" % name, DOC, flags=re.S).group(1)
    body = re.sub(r"//[^\n]*", " ", body)
    fields = []
    for decl in body.split(","):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"pub ([a-z0-9_]+): (.+)$", decl)
        assert m, decl
        fname, t = m.group(1), m.group(2).strip()
        count = 1
        am = re.match(r"\[([a-z0-9_]+); (\d+)\]$", t)
        if am:
            t, count = am.group(1), int(am.group(2))
        width = 8 if t.startswith("*") else {"u32": 4, "i32": 4, "f32": 4, "u64": 8, "f64": 8, "c_char": 1, "u8": 1}[t]
        fields.append((fname, width, count))
    return fields


def test_repr_c_structs_match_the_header_field_for_field():
    for name in ("gpe_config", "gpe_pipeline_info", "gpe_timing", "gpe_trace_event", "gpe_shard_plan", "gpe_shard_layout",
                 "gpe_shard_stats"):
        assert _c_struct_fields(name) == _rust_struct_fields(name), name


def test_rust_enum_and_constants_match_the_header():
    c_enum = re.search(r"typedef enum gpe_array \{(.*?)\} gpe_array;", _strip_c_comments(HEADER), flags=re.S).group(1)
    c_names = [re.sub(r"\s*=.*", "", x.strip()).replace("GPE_", "", 1) for x in c_enum.split(",") if x.strip()]
    r_enum = re.search(r"pub enum gpe_array \{(.*?)\}", DOC, flags=re.S).group(1)
    r_names = [re.sub(r"\s*=.*", "", x.strip()) for x in r_enum.split(",") if x.strip()]
    assert c_names == r_names
    for const in ("GPE_MODE_COMPAT", "GPE_MODE_NATIVE", "GPE_STEP_RESORT", "GPE_FLAG_NATIVE_FORCE", "GPE_FLAG_SORT_EVERY_STEP",
                  "GPE_FLAG_NATIVE_STATS", "GPE_FLAG_SAFE_SORT"):
        cv = re.search(r"\b%s\s*=\s*(\d+)" % const, HEADER).group(1)
        rv = re.search(r"pub const %s: u32 = (\d+);" % const, DOC).group(1)
        assert cv == rv, const


def test_the_three_host_mirrors_expose_the_same_particle_system_policy_surface():
    hpp = open(os.path.join(ROOT, "gpu-physics-engine_amd", "host", "gpe_host.hpp")).read()
    py = open(os.path.join(ROOT, "gpu-physics-engine_amd", "engine.py")).read()
    for method in ("is_it_time_to_sort", "reset_last_sort_time", "mouse_click_callback", "mouse_move_callback",
                   "sort_by_cell_id", "update_positions", "add_particles", "download_home_cell_ids",
                   "download_particle_ids", "download_particle_buffers", "get_max_radius"):
        assert re.search(r"pub fn %s\b" % method, DOC), "Rust shim text lacks %s" % method
        assert re.search(r"\b%s\s*\(" % method, hpp), "gpe_host.hpp lacks %s" % method
        assert re.search(r"def %s\b" % method, py), "engine.py lacks %s" % method
