#!/bin/bash
# the whole GPU suite + smoke, the way the driver runs them
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -x -q --timeout 1500 2>&1 | tail -15
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
