cd $GRAFT_REPO_ROOT
timeout 1100 python -m pytest tests -m gpu -q 2>&1 | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
