#!/bin/bash
# usage: gpu_kt.sh <label> <prof_run args...>   -- rocprofv3 kernel trace + stats of tools/prof_run.py, filtered summary on stdout
L=$1; shift
exec bash "$(dirname "$0")/gpu_kt_any.sh" "$L" tools/prof_run.py "$@"
