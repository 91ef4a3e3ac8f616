"""Regenerate tests/golden/script_records.json: what the REFERENCE's three example scripts hand to the
device when they run on the fedm_amd facade (build container only; the scripts are read from
/root/reference at run time and executed with their imports redirected, tests/script_harness.py).

    python tests/golden/make_script_records.py

The fixture holds numbers only: the bytes of the fedm_model_desc / fedm_gd_desc each script lowers
to, and per uploaded array its shape, sum, sum of squares and eight strided samples.
"""
import json
import sys
import tempfile
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
sys.path.insert(0, str(HERE.parent.parent))

import script_harness as sh     # noqa: E402

if __name__ == "__main__":
    records = {case: sh.digest(sh.run_reference_script(case, tempfile.mkdtemp(prefix=f"ref_{case}_")))
               for case in sh.SCRIPTS}
    (HERE / "script_records.json").write_text(json.dumps(records, separators=(",", ":")) + "\n")
    print({k: sorted(v) for k, v in records.items()})
