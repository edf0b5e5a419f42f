#!/usr/bin/env python3
"""Hostile input for the host layer's parsers (scene JSON, OBJ, PLY): mutated copies of the shipped files go through
cth_load_scene.  Any outcome but a memory error is fine -- a scene, or a refusal; run it against the sanitizer build
(tools/sanitize_host.sh), where a wild read or write ends the process with a report.
     python tools/fuzz_loader.py [seconds] [seed]"""
import os, sys, time, shutil, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import cobbletrace_amd as ca

A = os.path.join(os.path.dirname(__file__), "..", "assets")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
work = tempfile.mkdtemp(prefix="ct_fuzz_loader_")
os.makedirs(os.path.join(work, "models"))
tiny_ply = ("ply\nformat ascii 1.0\nelement vertex 5\nproperty float x\nproperty float y\nproperty float z\n"
            "element face 4\nproperty list uchar int vertex_indices\nend_header\n"
            "0 0 0\n1 0 0\n0 1 0\n0 0 1\n1 1 1\n3 0 1 2\n3 0 1 3\n3 1 2 4\n3 2 3 4\n").encode()
originals = {
    "scene.json": open(os.path.join(A, "scene_file_cube.json"), "rb").read(),
    "import.json": open(os.path.join(A, "scene_import.json"), "rb").read(),
    "plyscene.json": open(os.path.join(A, "scene_import_bunny.json"), "rb").read().replace(b"bunny.ply", b"tiny.ply"),
    "models/roundedCube.obj": open(os.path.join(A, "models", "roundedCube.obj"), "rb").read(),
    "models/tiny.ply": tiny_ply,
}
nasty = [b"-1", b"0", b"4294967295", b"2147483648", b"99999999999999999999", b"1e400", b"-1e-400", b"nan", b"inf", b"", b"{", b"}",
         b"[", b"]", b"\"", b",", b":", b"\x00", b"\n", b"//", b"3 0 1", b"f 1/2/3 4//5 6", b"f -1 -2 -3", b"element face 1000000"]

def mutate(data):
    b = bytearray(data)
    for _ in range(int(rng.integers(1, 4))):
        how, n = int(rng.integers(6)), len(b)
        if n == 0: break
        at = int(rng.integers(n))
        if how == 0: b[at] = int(rng.integers(256))
        elif how == 1: del b[at:at + int(rng.integers(1, 40))]
        elif how == 2: b[at:at] = b[at:at + int(rng.integers(1, 200))]
        elif how == 3: b = b[:at]
        elif how == 4: b[at:at] = nasty[int(rng.integers(len(nasty)))]
        else:                                               # replace the number under the cursor
            lo = at
            while lo > 0 and chr(b[lo - 1]) in "0123456789.-e": lo -= 1
            hi = at
            while hi < n and chr(b[hi]) in "0123456789.-e": hi += 1
            b[lo:hi] = nasty[int(rng.integers(len(nasty)))]
    return bytes(b)

def write_all(overrides):
    for name, data in originals.items():
        with open(os.path.join(work, name), "wb") as f: f.write(overrides.get(name, data))

from cobbletrace_amd import bindings
bindings.host_library()                                     # loaded once; every attempt runs in a forked child
err_path = os.path.join(work, "stderr.txt")

def attempt(path):
    """0: a scene; 1: refused (the parser's abort, the reference's assert); 2: anything else -- the sanitizer's report is printed"""
    pid = os.fork()
    if pid == 0:
        fd = os.open(err_path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC); os.dup2(fd, 2)
        try:
            s = ca.HostScene(path)
            d = s.desc
            _ = (d.n_triangles, d.n_nodes, float(np.asarray(s.vertices()).sum()) if d.n_triangles else 0.0)
            s.close()
        except BaseException:
            os._exit(3)
        os._exit(0)
    _, status = os.waitpid(pid, 0)
    if os.WIFEXITED(status) and os.WEXITSTATUS(status) in (0, 3): return 0 if os.WEXITSTATUS(status) == 0 else 1
    text = open(err_path, errors="replace").read()
    if "Sanitizer" in text or "runtime error" in text: print(text[-3000:]); return 2
    if os.WIFSIGNALED(status) and os.WTERMSIG(status) == 6 and "cobbletrace:" in text: return 1
    print("status %#x\n%s" % (status, text[-2000:])); return 2

counts = [0, 0, 0]
t_end, t_print = time.time() + budget, time.time() + 30
try:
    while time.time() < t_end and counts[2] == 0:
        victim = list(originals)[int(rng.integers(len(originals)))]
        mutant = mutate(originals[victim])
        write_all({victim: mutant})
        for scene in ("scene.json", "import.json", "plyscene.json"):
            if victim.endswith(".json") and victim != scene: continue
            if victim.endswith(".obj") and scene != "import.json": continue
            if victim.endswith(".ply") and scene != "plyscene.json": continue
            r = attempt(os.path.join(work, scene))
            counts[r] += 1
            if r == 2:
                keep = os.path.join(tempfile.gettempdir(), "ct_fuzz_loader_failure_" + os.path.basename(victim))
                open(keep, "wb").write(mutant); print("the input that did it:", keep)
        if time.time() > t_print:
            print("%d loaded, %d refused" % (counts[0], counts[1]), flush=True); t_print = time.time() + 30
finally:
    shutil.rmtree(work, ignore_errors=True)
print("%d mutated inputs: %d gave a scene, %d were refused (the parser's abort), %d broke the process" % (sum(counts), *counts))
sys.exit(1 if counts[2] else 0)
