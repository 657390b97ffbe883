#!/bin/bash
mkdir -p gpurun_out
tools/kb_run.sh t5i8_no t5i8_sa t5i8_sa3 t5i8_no_2 t5i8_sa_2 t5i8_sa3_2 p8i8_no p8i8_sa p8i8_sa3 t2i8_no t2i8_sa t2i8_sa3 tc5i8_no tc5i8_sa t5i8_no t5i8_sa p8i8_no p8i8_sa > gpurun_out/r03_kb20.txt 2>&1
grep -E "^==|cw/s" gpurun_out/r03_kb20.txt | paste - - | sed 's/| mean iters/it/; s/success.*hash/hash/; s/-DKIPT=1 -DKT=int8_t//' | cut -c1-230
