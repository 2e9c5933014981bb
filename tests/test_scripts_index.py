"""scripts/ holds only probes that run against the current library, and scripts/README.md says what each is."""
import os
import py_compile
import re

from conftest import ROOT

SCRIPTS = os.path.join(ROOT, "scripts")


def _tracked():
    return sorted(f for f in os.listdir(SCRIPTS) if f.endswith((".py", ".sh", ".hip")))


def test_readme_lists_exactly_the_scripts_that_exist():
    txt = open(os.path.join(SCRIPTS, "README.md")).read()
    listed = set(re.findall(r"`([A-Za-z0-9_]+\.(?:py|sh|hip))`", txt))
    assert listed == set(_tracked()), sorted(listed ^ set(_tracked()))


def test_every_python_script_compiles_and_names_only_living_kernel_forms(tmp_path):
    for f in _tracked():
        if f.endswith(".py"):
            py_compile.compile(os.path.join(SCRIPTS, f), cfile=str(tmp_path / (f + "c")), doraise=True)
            src = open(os.path.join(SCRIPTS, f)).read()
            for ln in src.splitlines():
                if "rng.choice([8, 9, 12" in ln or "FORMS = [" in ln:
                    nums = {int(x) for x in re.findall(r"\d+", ln.split("[", 1)[1])}
                    assert nums <= {0, 8, 9, 12, 13, 18}, (f, ln)
