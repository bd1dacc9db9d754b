"""oracle/README.md's clause-by-clause table (Appendix A statement -> oracle line -> product line) is generated from the statements' text
(scripts/make_oracle_readme.py): it must be up to date with the sources it cites."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_clause_table_is_current():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "make_oracle_readme.py"), "--check"], capture_output=True, text=True)
    assert res.returncode == 0, "oracle/README.md is stale: run python scripts/make_oracle_readme.py\n" + res.stdout + res.stderr
    text = open(os.path.join(ROOT, "oracle", "README.md")).read()
    assert text.count("| K") >= 25 and "Parity unpinned" in text
