#!/usr/bin/env python3
"""Extract the known-answer vectors held by the reference's own unit tests into a JSON fixture.

Run in the build container only (needs /root/reference); the output
`reference_kats.json` is committed and is what the tests read.  Only test DATA is
taken (expected values and the inputs they belong to), never source text:

  edge_crc / paritycheck_sum   src/codes/mod.rs:521-523 (CRC-32 of the edge stream), :109-241
  encode_parity                src/encoder.rs:361-527  (parity bytes for data = 0,1,2,...)
  hard_llr                     src/decoder.rs:553-605  (hard bytes <-> +-1 LLR pattern)
  sizes                        src/codes/mod.rs:109-241 CodeParams constants, asserted by
                               src/decoder.rs:531-551; table in src/lib.rs:176-188
  doctest_encode               src/lib.rs:135-143
"""
import json, re, sys, pathlib

REF = pathlib.Path("/root/reference")
CODES = ["TC128", "TC256", "TC512", "TM1280", "TM1536", "TM2048", "TM5120", "TM6144", "TM8192"]


def main():
    mod = (REF / "src/codes/mod.rs").read_text()
    enc = (REF / "src/encoder.rs").read_text()
    dec = (REF / "src/decoder.rs").read_text()

    kats = {"codes": CODES, "source": "adamgreig/labrador-ldpc v1.2.1 unit tests"}

    # CRC-32 known answers, in CODES order
    m = re.search(r"let crc_results = \[(.*?)\];", mod, re.S)
    kats["edge_crc"] = [int(x, 16) for x in re.findall(r"0x([0-9A-Fa-f]{8})", m.group(1))]
    assert len(kats["edge_crc"]) == 9

    # CodeParams constants
    sizes = {}
    for code in CODES:
        blk = re.search(r"pub const %s_PARAMS: CodeParams = CodeParams \{(.*?)\};" % code, mod, re.S).group(1)
        def field(name):
            expr = re.search(r"%s:\s*([^,]+)," % name, blk).group(1)
            return int(eval(expr.replace("/", "//")))
        sizes[code] = {f: field(f) for f in (
            "n", "k", "punctured_bits", "submatrix_size", "circulant_size", "paritycheck_sum",
            "decode_bf_working_len", "decode_ms_working_len", "decode_ms_working_u8_len", "output_len")}
    kats["sizes"] = sizes

    # encoder parity known answers
    par = {}
    for code in CODES:
        m = re.search(r"test_encode!\(LDPCCode::%s,\s*\[(.*?)\]\);" % code, enc, re.S)
        par[code] = [int(x, 16) for x in re.findall(r"0x([0-9A-Fa-f]{2})", m.group(1))]
        assert len(par[code]) * 8 == sizes[code]["n"] - sizes[code]["k"], code
    kats["encode_parity"] = par

    # hard <-> llr pattern (TC128)
    t = re.search(r"fn test_hard_to_llrs\(\)(.*?)fn test_llrs_to_hard", dec, re.S).group(1)
    hard = [int(x) for x in re.findall(r"\d+", re.search(r"let hard = vec!\[(.*?)\];", t, re.S).group(1))]
    pat = re.search(r"assert_eq!\(llrs, vec!\[(.*?)\]\);", t, re.S).group(1)
    llr_signs = [(-1 if tok.strip() == "llr" else 1) for tok in pat.replace("\n", " ").split(",") if tok.strip()]
    # `llr` is -1.0 in the test, `-llr` is +1.0
    assert len(hard) == 16 and len(llr_signs) == 128
    kats["hard_llr"] = {"code": "TC128", "hard": hard, "llrs": llr_signs}

    kats["doctest_encode"] = {
        "code": "TC128",
        "codeword": [0, 1, 2, 3, 4, 5, 6, 7, 0x34, 0x99, 0x98, 0x87, 0x94, 0xE1, 0x62, 0x56],
    }
    # scenario constants of test_decode_ms / test_decode_erasures / benches (not outputs):
    kats["decode_scenario"] = {"flip_byte0_mask": 0xA8, "maxiters": 50}

    out = pathlib.Path(__file__).with_name("reference_kats.json")
    out.write_text(json.dumps(kats, indent=1) + "\n")
    print("wrote", out)


if __name__ == "__main__":
    sys.exit(main())
