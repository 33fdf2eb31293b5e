"""Module constants the solve step reads (mirror of the reference's ``src/parameters.py``:
``BA_WINDOW_SIZE`` :19, ``DEBUG_DIRS['lba_steps']`` :16 used at
``src/bundle_adjuster.py:187``)."""
OUTPUT_DIR = 'output_map'
DEBUG = True
DEBUG_DIRS = {
    'sparsity': 'debug_sparsity',
    'lba_steps': 'output_map/lba_steps',
}
BA_WINDOW_SIZE = 5
