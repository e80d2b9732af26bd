"""Print a bench.py JSON line compactly:  python tools/show_bench.py gpurun_out/x/bench.json"""
import json, sys
for path in sys.argv[1:]:
    t = open(path).read().strip()
    if not t:
        print(path, "EMPTY"); continue
    line = t.splitlines()[-1]
    if line.startswith("INPROC "):
        line = line[7:]
    d = json.loads(line)
    print("==", path)
    for k, v in d.items():
        if isinstance(v, dict):
            print(" ", k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a not in ("timing", "recipe", "sample", "workload", "parallelism")})
            for a in ("sample", "parallelism"):
                if a in v: print("     ", a + ":", v[a])
        elif k not in ("metric", "unit", "higher_is_better", "vs_baseline", "dtype", "data"):
            print(" ", k, v)
