"""CPU: the proof parser + verifier (host-only product code) under AddressSanitizer / UBSan, fed with thousands of mutated
wire images.  GPU sanitizers are unavailable on the pool, so the memory-safety check of the untrusted-input surface
(frieda_proof_deserialize, frieda_verify) runs here on the CPU build of the same sources."""
import os
import subprocess

import pytest

from conftest import ROOT, pattern_bytes

CSRC = os.path.join(ROOT, "frieda_amd", "csrc")
HOST_SOURCES = ["proof.cpp", "verifier.cpp", "transcript.cpp", "context.cpp", "capi_host_stub.cpp"]


@pytest.fixture(scope="module")
def fuzz_binary(tmp_path_factory):
    out = tmp_path_factory.mktemp("asan")
    stub = out / "capi_host_stub.cpp"
    # the slice of the C ABI the fuzzer needs, without the device-side entry points (which need hipcc)
    stub.write_text(
        '#include <string.h>\n#include <new>\n#include "host.h"\nusing namespace frieda;\n'
        'namespace frieda { namespace k { void gen_twiddles(const Launch&, uint32_t, const TwiddleSeeds&, uint32_t*, uint32_t*, void*) {} } '
        "void ProveJobDeleter::operator()(ProveJob*) const {} }\n"
        'extern "C" {\n'
        "int frieda_proof_deserialize(const uint8_t* buf, size_t len, frieda_proof** out) { if (!buf || !out) return 1; *out = nullptr; "
        "frieda_proof* p = new frieda_proof(); if (!deserialize_proof(buf, len, p->p)) { delete p; return 5; } *out = p; return 0; }\n"
        "int frieda_verify(const frieda_proof* p, const uint64_t* seed, int* ok) { try { return verify(p->p, seed, ok); } catch (...) { return 3; } }\n"
        "void frieda_proof_free(frieda_proof* p) { delete p; }\n}\n"
    )
    exe = out / "fuzz_verify"
    srcs = [os.path.join(CSRC, s) for s in HOST_SOURCES[:-1]] + [str(stub), os.path.join(ROOT, "tests", "cpp", "fuzz_verify.cpp")]
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-D__HIP_PLATFORM_AMD__",
           "-I/opt/rocm/include", "-I" + CSRC, "-I" + os.path.join(ROOT, "include"), "-Wno-unused-parameter", *srcs, "-o", str(exe),
           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        # only a missing sanitizer runtime is a reason to skip; a source that no longer builds or links on the host is a failure
        # (round 5: a new cross-file reference in context.cpp silently turned this test into a skip)
        if "asan" in r.stderr.lower() and ("cannot find" in r.stderr or "No such file" in r.stderr):
            pytest.skip("host sanitizer runtime unavailable: " + r.stderr[-400:])
        pytest.fail("host sanitizer build failed: " + r.stderr[-1500:])
    return str(exe)


@pytest.mark.parametrize("case", ["pattern", "blob"])
def test_parser_and_verifier_under_asan(fuzz_binary, oracle, blob, tmp_path, case):
    if case == "pattern":
        data, seed, cfg = pattern_bytes(1024).tobytes(), 1024, oracle.make_config(12, 4, 0, 20)
    else:
        data, seed, cfg = blob, None, oracle.make_config(12, 4, 1, 20)
    _, proof = oracle.commit_and_generate_proof(data, seed, cfg)
    img = tmp_path / "proof.bin"
    img.write_bytes(proof.serialize())
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([fuzz_binary, str(img), "-" if seed is None else str(seed), "3000"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert "accepted_mutants 0" in r.stdout


def test_lazy_field_arithmetic_matches_textbook(tmp_path):
    """field.h's lazily reduced fold (one 64-bit accumulator per coordinate) equals the textbook QM31 sequence, under UBSan,
    including the operands that attain the accumulator bound."""
    exe = tmp_path / "test_field"
    cmd = ["g++", "-std=c++17", "-O2", "-fsanitize=undefined", "-fno-sanitize-recover=undefined", "-I" + CSRC,
           os.path.join(ROOT, "tests", "cpp", "test_field.cpp"), "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-800:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr
