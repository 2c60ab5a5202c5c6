"""The files under profiles/ are evidence only if they agree with each other and with the documents that quote them (the
round-4 review's first item).  CPU-only: the kept kernel statistics must reproduce the bench line kept beside them
(tools/check_profile_consistency.py, the check tools/collect_profiles.sh ends with), and the generated parts of README.md,
DESIGN.md and profiles/README.md must be what tools/refresh_readme.py makes of the kept files."""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest_round():
    reports = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_report.json")))
    assert reports, "no kept bench report"
    return os.path.basename(reports[-1])[:3]


def test_kept_kernel_statistics_reproduce_the_kept_line():
    tag = _newest_round()
    for line, stats in (("headline.json", "headline_kernel_stats.csv"), ("lane_headline.json", "lane_headline_kernel_stats.csv")):
        a, b = os.path.join(ROOT, "profiles", "%s_%s" % (tag, line)), os.path.join(ROOT, "profiles", "%s_%s" % (tag, stats))
        if not (os.path.exists(a) and os.path.exists(b)):
            continue
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_profile_consistency.py"), a, b],
                             capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, out.stdout + out.stderr


def test_generated_documents_are_what_the_kept_files_give(tmp_path):
    """tools/refresh_readme.py on a copy of the tree must change nothing."""
    copy = tmp_path / "repo"
    for sub in ("profiles", "tools"):
        shutil.copytree(os.path.join(ROOT, sub), copy / sub, ignore=shutil.ignore_patterns("*.so", "__pycache__"))
    for f in ("README.md", "DESIGN.md"):
        shutil.copy(os.path.join(ROOT, f), copy / f)
    out = subprocess.run([sys.executable, str(copy / "tools" / "refresh_readme.py")], capture_output=True, text=True, timeout=120, cwd=str(copy))
    assert out.returncode == 0, out.stdout + out.stderr
    for f in ("README.md", "DESIGN.md", os.path.join("profiles", "README.md")):
        assert open(os.path.join(ROOT, f)).read() == open(copy / f).read(), "%s is not what tools/refresh_readme.py generates from profiles/" % f
