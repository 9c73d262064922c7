"""`python main.py ...` like the reference (main.py); the implementation lives in pcrlv2_amd/main.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pcrlv2_amd.main import main  # noqa: E402

if __name__ == '__main__':
    main()
