#!/bin/bash
# On the GPU box (whose tree is a scratch copy): put the library named first in place of
# raytracer_amd/librtx_hip.so, then run the rest of the command line.
set -e
so=$1; shift
cp "$so" raytracer_amd/librtx_hip.so
exec "$@"
