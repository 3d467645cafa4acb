"""Build the HIP shared library in-tree (gfx950 only).

Every source is compiled to its own object under leanmultisig_amd/_obj/ (git-ignored) and only re-compiled when it or a
header it may include has changed; objects are compiled in parallel and linked into libleanmultisig_hip.so."""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libleanmultisig_hip.so")
SOURCES = ["lm_core.hip", "lm_commit.hip", "lm_whir_ops.hip", "lm_gkr.hip", "lm_air.hip", "lm_logup.hip", "lm_vm_device.hip", "host/lm_host.cpp",
           "host/lm_whir_config.cpp", "host/lm_wire.cpp", "host/lm_verify.cpp", "host/lm_poseidon_x86.cpp", "host/lm_vm.cpp",
           "host/lm_node.cpp", "host/lm_aggregate.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
if os.environ.get("LM_PUBLISH_FENCES"):  # conservative hand-over with release fences (csrc/lm_common.h)
    FLAGS.append("-DLM_PUBLISH_FENCES=1")
# The conservative variant (release fences before every flag / ticket instead of the fence-free hand-over of lm_common.h) is built
# next to the product library so that the parity tests can run on it (tests/test_fences_gpu.py: LM_LIB selects the library).
FENCES_OBJ = os.path.join(HERE, "_obj_fences")
FENCES_LIB = os.path.join(HERE, "libleanmultisig_hip_fences.so")


def _sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _headers():
    deps = []
    for root, _, files in os.walk(CSRC):
        for f in files:
            if f.endswith((".h", ".inc", ".hpp")):
                deps.append(os.path.join(root, f))
    deps.append(os.path.join(os.path.dirname(HERE), "include", "leanmultisig.h"))
    deps.append(os.path.join(os.path.dirname(HERE), "include", "leanmultisig_host.h"))
    return sorted(deps)


def _includes(src, headers_by_name, seen):
    """transitive closure of the quoted includes of `src` among this tree's headers (file names are unique)"""
    try:
        text = open(src, errors="replace").read()
    except OSError:
        return
    for line in text.splitlines():
        line = line.strip()
        if line.startswith("#include") and '"' in line:
            name = os.path.basename(line.split('"')[1])
            h = headers_by_name.get(name)
            if h and h not in seen:
                seen.add(h)
                _includes(h, headers_by_name, seen)


def _stamp(src, headers_by_name, flags=None):
    h = hashlib.sha256()
    h.update(" ".join(flags or FLAGS).encode())
    seen = set()
    _includes(src, headers_by_name, seen)
    for f in [src] + sorted(seen):
        h.update(f.encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _obj_path(src, obj_dir=None):
    rel = os.path.relpath(src, CSRC).replace(os.sep, "_")
    return os.path.join(obj_dir or OBJ, rel + ".o")


def needs_build(flags=None, obj_dir=None, lib=None):
    lib = lib or LIB
    if not os.path.exists(lib):
        return True
    headers_by_name = {os.path.basename(h): h for h in _headers()}
    for s in _sources():
        o = _obj_path(s, obj_dir)
        if not os.path.exists(o) or not os.path.exists(o + ".sha") or open(o + ".sha").read() != _stamp(s, headers_by_name, flags):
            return True
    return os.path.getmtime(lib) < max(os.path.getmtime(_obj_path(s, obj_dir)) for s in _sources())


def build(force=False, verbose=True, variants=True):
    """hipcc --offload-arch=gfx950 -> leanmultisig_amd/libleanmultisig_hip.so (+ the release-fence variant libleanmultisig_hip_fences.so)"""
    consts = os.path.join(CSRC, "poseidon16_consts.inc")
    gen = os.path.join(CSRC, "gen_poseidon_consts.py")
    if not os.path.exists(consts) or os.path.getmtime(gen) > os.path.getmtime(consts):
        subprocess.check_call([sys.executable, gen])
    _build_one(FLAGS, OBJ, LIB, force, verbose)
    if variants and "-DLM_PUBLISH_FENCES=1" not in FLAGS:
        _build_one(FLAGS + ["-DLM_PUBLISH_FENCES=1"], FENCES_OBJ, FENCES_LIB, force, verbose)
    return LIB


def _build_one(flags, obj_dir, lib, force, verbose):
    if not force and not needs_build(flags, obj_dir, lib):
        return lib
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers_by_name = {os.path.basename(h): h for h in _headers()}
    todo = []
    for s in _sources():
        o, st = _obj_path(s, obj_dir), _stamp(s, headers_by_name, flags)
        if force or not os.path.exists(o) or not os.path.exists(o + ".sha") or open(o + ".sha").read() != st:
            todo.append((s, o, st))

    def compile_one(job):
        s, o, st = job
        cmd = [hipcc, *flags, "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(o + ".sha", "w") as f:
            f.write(st)

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(compile_one, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *[_obj_path(s, obj_dir) for s in _sources()], "-o", lib, "-lpthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv)
