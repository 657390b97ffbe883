"""The Rust host binding (bindings/rust/) cannot be compiled in this image (no rustc / cargo), so it is checked mechanically:
every function include/labrador_ldpc_hip.h declares must appear in the shim's `extern "C"` block with the same name, the same
number of arguments and matching argument / return types, the #[repr(C)] struct must list the header's fields in order, and
the ABI constant must be the header's.  What the shim mirrors: /root/reference/capi/src/lib.rs:15-179, in the opposite
direction (round 2's review, item 8)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "labrador_ldpc_hip.h")
SHIM = os.path.join(ROOT, "bindings", "rust", "src", "lib.rs")

C_TO_RUST = {"size_t": "usize", "int": "c_int", "bool": "bool", "void": "()", "float": "f32", "double": "f64",
             "uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64", "int8_t": "i8", "int16_t": "i16",
             "int32_t": "i32", "char": "c_char", "enum labrador_ldpc_code": "LDPCCode", "struct labrador_ldpc_hip_opts": "HipOpts"}


def c_type_to_rust(t):
    """`const float *const *` -> `*const *const f32`: every `*` points to what stands left of it, const or not."""
    segs = [" ".join(x.split()) for x in t.split("*")]       # ["const float", "const", ""]: base, then the qualifier behind each star
    base = segs[0]
    const = base.startswith("const ")
    if const:
        base = base[6:]
    r = C_TO_RUST[base]
    if len(segs) == 1:
        return r
    if base == "void":
        r = "c_void"
    for q in segs[1:]:                                      # one pointer level per star, inside out
        r = ("*const " if const else "*mut ") + r
        const = q == "const"
    return r


def header_functions():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    src = "\n".join(l for l in src.split("\n") if not l.strip().startswith("#"))
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w \*]*?)\b(labrador_ldpc_\w+)\s*\(([^;{}]*?)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?)(\w+)$", a)            # last identifier is the parameter name
                params.append(c_type_to_rust(mm.group(1).strip()))
        out[name] = (c_type_to_rust(ret), params)
    return out


def shim_functions():
    src = open(SHIM).read()
    block = src[src.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    out = {}
    for m in re.finditer(r"pub fn (labrador_ldpc_\w+)\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        name, args, ret = m.group(1), m.group(2).strip(), (m.group(3) or "()").strip()
        params = [a.split(":", 1)[1].strip() for a in args.split(",") if a.strip()]
        out[name] = (ret, params)
    return out


def test_every_header_function_is_bound_with_the_same_signature():
    h, r = header_functions(), shim_functions()
    assert len(h) >= 49, f"the header parser found only {len(h)} functions"
    assert set(h) == set(r), f"missing in the shim: {sorted(set(h) - set(r))}; unknown to the header: {sorted(set(r) - set(h))}"
    for name, (ret, params) in h.items():
        assert r[name][0] == ret, f"{name}: returns {r[name][0]} in the shim, {ret} in the header"
        assert r[name][1] == params, f"{name}: arguments {r[name][1]} in the shim, {params} in the header"


def test_the_reference_symbols_are_all_there():
    """The 21 symbols of the reference's C API (capi/src/lib.rs:15-179; listed in tests/golden/reference_kats.json's source)."""
    ref = ["code_n", "code_k", "bf_working_len", "ms_working_len", "ms_working_u8_len", "output_len", "encode", "copy_encode",
           "decode_bf", "decode_ms_i8", "decode_ms_i16", "decode_ms_f32", "decode_ms_f64"] + \
          [f"{f}_{t}" for f in ("hard_to_llrs", "llrs_to_hard") for t in ("i8", "i16", "f32", "f64")]
    assert len(ref) == 21
    r = shim_functions()
    for s in ref:
        assert "labrador_ldpc_" + s in r


def test_opts_struct_and_abi_match_the_header():
    h = open(HEADER).read()
    abi = int(re.search(r"#define LABRADOR_LDPC_HIP_ABI (\d+)", h).group(1))
    body = re.sub(r"/\*.*?\*/", "", h[h.index("struct labrador_ldpc_hip_opts {"):], flags=re.S)
    body = body[body.index("{") + 1:body.index("};")]
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            mm = re.match(r"(.*?)(\w+)$", decl)
            fields.append((mm.group(2), c_type_to_rust(mm.group(1).strip())))
    s = open(SHIM).read()
    assert int(re.search(r"pub const ABI: c_int = (\d+);", s).group(1)) == abi
    sb = s[s.index("pub struct HipOpts {"):]
    sb = sb[sb.index("{") + 1:sb.index("}")]
    rust_fields = [(m.group(1), m.group(2).strip()) for m in re.finditer(r"pub (\w+): ([^,]+),", sb)]
    assert rust_fields == fields, (rust_fields, fields)
    assert "#[repr(C)]" in s[:s.index("pub struct HipOpts {")].rsplit("\n\n", 1)[-1]
    # constants
    for name, rust in (("LABRADOR_LDPC_HIP_MEM_DEVICE", "MEM_DEVICE"), ("LABRADOR_LDPC_HIP_DEVICE_ALL", "DEVICE_ALL"),
                       ("LABRADOR_LDPC_HIP_DEVICE_CURRENT", "DEVICE_CURRENT"), ("LABRADOR_LDPC_HIP_EUNSUPPORTED", "EUNSUPPORTED"),
                       ("LABRADOR_LDPC_HIP_ENODEV", "ENODEV")):
        hv = int(re.search(rf"#define {name}\s+\(?(-?\d+)\)?", h).group(1))
        rv = int(re.search(rf"pub const {rust}: c_int = (-?\d+);", s).group(1))
        assert hv == rv, name
    # the enum's discriminants are the reference's (src/codes/mod.rs:37-66)
    names = ["TC128", "TC256", "TC512", "TM1280", "TM1536", "TM2048", "TM5120", "TM6144", "TM8192"]
    for i, n in enumerate(names):
        assert re.search(rf"\b{n} = {i},", s)
