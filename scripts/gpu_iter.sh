#!/bin/bash
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "^E  |passed|failed|^FAILED|Error" | head -20
timeout 300 python scripts/bench_match.py 2>/dev/null | cut -c1-300
