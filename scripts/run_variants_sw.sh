#!/bin/bash
for v in "$@"; do echo "== $v"; ISOCON_LIB=$PWD/isocon_amd/lib/$v python scripts/dev/quick_sw.py 2>&1 | tail -2 | head -1; done
