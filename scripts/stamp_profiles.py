"""After copying a round's PMC summaries from gpurun_out/<rNN>/ into profiles/: record the commit they were collected at in
each JSON's `_meta` (the GPU box has no .git; bench.py prints it next to every field it reads from these files).
    python scripts/stamp_profiles.py r05 [commit]"""
import json
import subprocess
import sys
from pathlib import Path

root = Path(__file__).resolve().parent.parent
rnd = sys.argv[1]
commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=root, text=True).strip()
for f in sorted((root / "profiles").glob(f"{rnd}_*pmc_*.json")):
    d = json.loads(f.read_text())
    d["_meta"] = {"collected_at_commit": commit, "round": rnd, "command": "scripts/collect_profiles.sh " + rnd}
    f.write_text(json.dumps(d, indent=1, sort_keys=True))
    print(f.name, commit)
