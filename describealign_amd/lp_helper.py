"""Helper process of lp_tree.solve_parallel:  python -m describealign_amd.lp_helper

Reads length-prefixed pickled sub-LP tasks (the arguments of lp_tree._pool_solve) from stdin, writes the pickled results to stdout.
Started with subprocess by lp_tree.HelperPool -- not with multiprocessing: a spawned multiprocessing child re-imports the parent's
__main__ module, and align() must stay callable from a script without a `if __name__ == "__main__"` guard.  Never touches a GPU."""
import pickle
import struct
import sys


def main():
  from describealign_amd import lp_tree
  inp, out = sys.stdin.buffer, sys.stdout.buffer
  ok = lp_tree.available()
  out.write(b"R" if ok else b"N"); out.flush()
  while True:
    head = inp.read(8)
    if len(head) < 8:
      return
    (size,) = struct.unpack("<Q", head)
    args = pickle.loads(inp.read(size))
    try:
      res = lp_tree._pool_solve(*args)
    except Exception as e:                       # noqa: BLE001 -- reported to the parent, which falls back
      res = ("error", f"{type(e).__name__}: {e}")
    blob = pickle.dumps(res, protocol=pickle.HIGHEST_PROTOCOL)
    out.write(struct.pack("<Q", len(blob))); out.write(blob); out.flush()


if __name__ == "__main__":
  main()
