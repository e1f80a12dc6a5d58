#!/bin/bash
python -c "import __graft_entry__ as g; g.smoke()"
