"""DWT modules and autograd Functions of the MI355X engine (see transform2d.py, lowlevel.py)."""
