#!/bin/bash
# CPU suite with the full log kept (flaky-failure forensics): tools/cpu_tests.sh
cd "$(dirname "$0")/.." && python -m pytest tests -x -q -m "not gpu" -rf > /tmp/dgs_cpu_tests.log 2>&1
rc=$?
tail -1 /tmp/dgs_cpu_tests.log
if [ $rc -ne 0 ]; then cp /tmp/dgs_cpu_tests.log /tmp/dgs_cpu_tests_failed_$(date +%s).log; grep -E "^(E |FAILED)" /tmp/dgs_cpu_tests.log | head -20; fi
exit $rc
