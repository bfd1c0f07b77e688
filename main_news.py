"""SML on Adressa: 63 periods, online training from period 21, testing from period 48.
Same command line as the reference's main_news.py; see sml_amd/cli.py."""
from sml_amd.cli import get_parse as _gp, main


def get_parse():
    return _gp("news")


if __name__ == "__main__":
    main("news")
