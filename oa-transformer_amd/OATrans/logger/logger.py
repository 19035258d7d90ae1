"""Console + rotating-file logging (stand-in for /root/reference/OATrans/logger/logger.py:7-22)."""
import logging
import logging.config
from pathlib import Path


def setup_logging(save_dir, default_level=logging.INFO):
    save_dir = Path(save_dir)
    logging.config.dictConfig({
        "version": 1, "disable_existing_loggers": False,
        "formatters": {"simple": {"format": "%(message)s"},
                       "datetime": {"format": "%(asctime)s - %(name)s - %(levelname)s - %(message)s"}},
        "handlers": {
            "console": {"class": "logging.StreamHandler", "level": "DEBUG", "formatter": "simple",
                        "stream": "ext://sys.stdout"},
            "info_file_handler": {"class": "logging.handlers.RotatingFileHandler", "level": "INFO",
                                  "formatter": "datetime", "filename": str(save_dir / "info.log"),
                                  "maxBytes": 10485760, "backupCount": 20, "encoding": "utf8"}},
        "root": {"level": "INFO", "handlers": ["console", "info_file_handler"]}})
