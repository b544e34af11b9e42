"""A tiny job launcher for the GPU tests.  conftest.py starts it ONCE, before anything in the pytest process has
touched the GPU; it never touches the GPU itself.  A test that needs a fresh GPU process tree (the data-parallel
trainer under `python -m torch.distributed.run`) sends a JSON line {"cmd": [...], "env": {...}, "timeout": s} and
reads back {"rc", "out", "err"}: the job is a child of THIS process, so nothing that has initialised the GPU ever
forks or execs."""
import json
import os
import subprocess
import sys


def main():
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        job = json.loads(line)
        if job.get("quit"):
            break
        env = dict(os.environ)
        env.update(job.get("env", {}))
        try:
            r = subprocess.run(job["cmd"], env=env, capture_output=True, text=True, timeout=job.get("timeout", 600), cwd=job.get("cwd"))
            res = {"rc": r.returncode, "out": r.stdout[-20000:], "err": r.stderr[-20000:]}
        except subprocess.TimeoutExpired as e:
            res = {"rc": -9, "out": (e.stdout or b"").decode(errors="replace")[-20000:] if isinstance(e.stdout, bytes) else (e.stdout or ""),
                   "err": "timeout"}
        sys.stdout.write(json.dumps(res) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
