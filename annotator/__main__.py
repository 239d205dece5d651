import sys

from dnncancerannotator_amd.__main__ import main

if __name__ == '__main__':
    sys.exit(main(prog='python3 -m annotator'))
