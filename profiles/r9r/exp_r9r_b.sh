#!/bin/bash
out=gpurun_out/r9r; mkdir -p $out
for form in ${FORMS:-4 5 1 0}; do PEANUT_SKINNY_FORM=$form python profiles/r9r/probe.py 20 2>&1 | grep "^form" | python -c "
import sys, ast
for l in sys.stdin:
    head, d = l.split(': ', 1)
    d = ast.literal_eval(d)
    print(head, 'distinct hashes', len(d), 'of', sum(d.values()), 'first', sorted(d)[0])
"; done | tee $out/forms_b.txt
