"""one build of the library (RELATE_AMD_LIB) on dumped matrices: phase times of the tree-build kernel per tree"""
import os, re, subprocess, sys
import numpy as np
if len(sys.argv) > 2 and sys.argv[2] == "child":
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    from relate_amd import api
    d = sys.argv[1]
    ks = sorted(int(f[2:-4]) for f in os.listdir(d) if f.startswith("d_"))
    N = int(round((os.path.getsize(os.path.join(d, "d_%d.bin" % ks[0])) / 4) ** 0.5))
    b = api.Builder(N, device=0)
    out = []
    for rep in range(2):
        for k in ks:
            dm = np.fromfile(os.path.join(d, "d_%d.bin" % k), np.float32).reshape(N, N)
            cf = os.path.join(d, "cf_%d.bin" % k)
            pr = np.fromfile(cf, np.float32).reshape(N, N) if os.path.exists(cf) else None
            out.append(b.build(dm, pr)[0])
    b.close()
    import hashlib
    print("parents md5", hashlib.md5(b"".join(x.tobytes() for x in out)).hexdigest())
    sys.exit(0)
p = subprocess.run([sys.executable, __file__, sys.argv[1], "child"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                   env=dict(os.environ, RELATE_AMD_TIMING="1"), timeout=300)
acc, n = {}, 0
for l in p.stderr.decode().split("\n"):
    if "[gpu tree builder]" in l and "us:" in l:
        n += 1
        for m in re.finditer(r"([a-z_+ ]+?) (\d+)(?= |$)", l.split("us:")[1]):
            acc[m.group(1).strip()] = acc.get(m.group(1).strip(), 0) + int(m.group(2))
tot = sum(v for k, v in acc.items() if k not in ("pairs_x100", "shader_MHz", "z"))
raw = [l for l in p.stderr.decode().split("\n") if "[gpu tree builder]" in l and "us:" in l]
if os.environ.get("MM_RAW") and raw:
    print(raw[-1])
print(os.path.basename(os.environ.get("RELATE_AMD_LIB", "default")), "rc", p.returncode, "trees", n,
      "ms/tree %.1f" % (tot / max(n, 1) / 1000.0), {k: round(v / max(n, 1) / 1000.0, 1) for k, v in acc.items()},
      p.stdout.decode().strip()[-60:], p.stderr.decode()[-200:] if p.returncode else "")
