#!/bin/bash
# The GPU suite from a CLEAN COPY of the tracked files only (what `git clone` gives: no built libraries, no MIOpen caches, no gpurun_out) on a
# fresh lease: build on the box, pytest -m gpu -x, smoke.  The GPU box has no .git, so the list of tracked files travels as
# .tracked_files.txt (written by `git ls-files > .tracked_files.txt` before the gpurun call; untracked itself).
set -e
rm -rf /tmp/clean && mkdir -p /tmp/clean
tar cf - -T .tracked_files.txt | tar xf - -C /tmp/clean
cd /tmp/clean
echo "clean copy: $(find . -type f | wc -l) tracked files, built libraries in it: $(find . -name '*.so' | wc -l)"
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=12 2>&1 | tail -40
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
