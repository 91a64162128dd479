"""Run bench.py against another build of the library (experiments: python tools/bench_with_lib.py <lib.so> [bench args])."""
import pathlib, runpy, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import demethify_amd._lib as L
L.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path(str(pathlib.Path(__file__).resolve().parent.parent / "bench.py"), run_name="__main__")
