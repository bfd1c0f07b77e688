"""SML on Yelp: 40 periods, online training from period 10, testing from period 30.
Same command line as the reference's main_yelp.py; see sml_amd/cli.py."""
from sml_amd.cli import get_parse as _gp, main


def get_parse():
    return _gp("yelp")


if __name__ == "__main__":
    main("yelp")
