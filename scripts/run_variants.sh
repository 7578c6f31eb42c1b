#!/bin/bash
for v in "$@"; do ISOCON_LIB=$PWD/isocon_amd/lib/$v python scripts/dev/quick_scan.py 2>&1 | tail -1; done
python scripts/dev/quick_scan.py 2>&1 | tail -1
