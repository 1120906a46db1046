import sys

from .cli import main

if __name__ == "__main__":
    rc = main()
    sys.exit(rc if isinstance(rc, int) else 0)
