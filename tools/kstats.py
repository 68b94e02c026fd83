"""print a rocprofv3 kernel_stats.csv compactly: name prefix, calls, avg/min/max us"""
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    print('%-44s calls %5s  avg %9.1f us  min %9.1f  max %9.1f  %5s%%' % (row['Name'][:44], row['Calls'], float(row['AverageNs']) / 1e3,
          float(row['MinNs']) / 1e3, float(row['MaxNs']) / 1e3, row['Percentage']))
