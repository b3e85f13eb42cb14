#!/bin/bash
micro() { python bench.py "$@" --micro-only 2>&1 | grep "^micro" | cut -c1-200; }
for o in hyperplane natural; do echo "== cell order $o"; micro --config c4 --cell-order $o; micro --config c5 --cell-order $o; micro --cell-order $o; micro --rank-share 8 --cell-order $o; done
