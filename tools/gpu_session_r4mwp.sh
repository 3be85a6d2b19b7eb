#!/bin/bash
# round 4: where the Ant's one launch spends its time (instrumented library, tools/debug/mw_phases.py)
out=gpurun_out/r4mwp; mkdir -p $out
MI_ENGINE_LIB=$PWD/ab/lib_timing_mw.so timeout 300 python tools/debug/mw_phases.py > $out/ant_mw_phases.txt 2> $out/err.log; echo "rc=$?"; cat $out/ant_mw_phases.txt; tail -5 $out/err.log
