"""`from utils import utils_logger` (test_demo.py:9): logger_info(name, log_path) with the reference's line format."""
import logging


def logger_info(logger_name, log_path='default_logger.log'):
    log = logging.getLogger(logger_name)
    if log.hasHandlers():
        print('LogHandlers exist!')
        return
    print('LogHandlers setup!')
    fmt = logging.Formatter('%(asctime)s.%(msecs)03d : %(message)s', datefmt='%y-%m-%d %H:%M:%S')
    log.setLevel(logging.INFO)
    for h in (logging.FileHandler(log_path, mode='a'), logging.StreamHandler()):
        h.setFormatter(fmt)
        log.addHandler(h)
