#!/bin/bash
out=gpurun_out/r9r; mkdir -p $out
for form in 0 1 2 3 1 0; do PEANUT_SKINNY_FORM=$form python profiles/r9r/probe.py 30 2>&1 | grep "^form"; done | tee $out/forms.txt
