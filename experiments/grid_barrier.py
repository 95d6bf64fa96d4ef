"""Runs the device-wide barrier timing (experiments/grid_barrier.hip)."""
import ctypes, os, sys
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgridbarrier.so"))
sys.exit(lib.grid_barrier_run())
