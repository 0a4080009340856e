"""`install()` makes `import assistive_gym` resolve to the drop-in package in this directory."""
import os
import sys


def install():
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    return here
