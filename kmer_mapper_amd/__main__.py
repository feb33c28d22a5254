"""`python -m kmer_mapper_amd map ...` == the reference's `kmer_mapper map ...` console script (setup.py:31-33)."""
from .command_line_interface import main

if __name__ == "__main__":
    main()
