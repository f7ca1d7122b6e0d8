"""include/zvec_hip_operator.hpp without a GPU: the micro-batcher, the probe-parameter arithmetic and the key directory are plain
host C++ (tests/cpp/test_batcher_cpu.cc: 48 caller threads over a stand-in batched search); and the two host sides that instantiate
the operator templates — the C++ mirror's test program and the load tool — compile against the header."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_micro_batcher_probe_params_and_key_directory_on_cpu():
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "test_batcher_cpu")
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_batcher_cpu.cc"),
                               "-L" + os.path.join(ROOT, "zvec_amd"), "-lzvec_hip", "-Wl,-rpath," + os.path.join(ROOT, "zvec_amd")])
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stdout + out.stderr
        assert out.stdout.strip().endswith("ok")


def test_both_host_sides_instantiate_the_shared_operator_templates():
    inc = ["-I" + os.path.join(ROOT, "include")]
    for src in (os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cc"), os.path.join(ROOT, "tools", "cpp", "load_bench.cc")):
        subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wno-comment"] + inc + [src])
    # the real plugin, when the reference's headers are here (build container only)
    ref = "/root/reference/src"
    if os.path.isdir(os.path.join(ref, "include", "zvec")):
        for tu in ("hip_plugin.cc", "hip_ivf_builder.cc"):
            subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-w", "-I" + os.path.join(ref, "include"), "-I" + ref,
                                   "-I" + os.path.join(ref, "core"), "-I" + os.path.join(ref, "core", "algorithm")] + inc +
                                  ["-I" + os.path.join(ROOT, "plugin"), os.path.join(ROOT, "plugin", tu)])
