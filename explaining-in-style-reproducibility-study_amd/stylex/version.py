__version__ = '1.8.7'
