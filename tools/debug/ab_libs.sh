for lib in "$@"; do echo "== $lib"; MI_ENGINE_LIB=$PWD/$lib python tools/mwc_ab.py 2>&1 | grep "multi_wave=32"; done
